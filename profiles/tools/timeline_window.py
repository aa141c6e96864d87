"""Prints every kernel of a window in the middle of a rocprofv3 kernel trace (csv): start
offset, duration, queue and name; for looking at how the streams of a run interleave.
    python profiles/tools/timeline_window.py <trace dir> [kernels to print]"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
middle = len(rows) // 2
t0 = int(rows[middle]["Start_Timestamp"])
for r in rows[middle:middle + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  dur %8.1f us  queue %-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
