"""Prints every kernel of the last ~3 frames of a rocprofv3 kernel trace (csv) with stream
(queue) ids: start offset, duration in microseconds."""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
shades = [i for i, r in enumerate(rows) if "shade_pixels" in r["Kernel_Name"]]
first = shades[-4]
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:shades[-1] + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  dur %8.1f us  queue %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-48:]))
