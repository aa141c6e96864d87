"""Prints the kernel timeline of one steady-state frame from a rocprofv3 kernel trace
(csv): start offset, duration and gap to the previous kernel in microseconds."""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
# last frame = from the last shade_pixels on
last = max(i for i, r in enumerate(rows) if "shade_pixels" in r["Kernel_Name"])
first = max(i for i, r in enumerate(rows[:last]) if "shade_pixels" in r["Kernel_Name"])
t0 = int(rows[first]["Start_Timestamp"])
previous_end = None
for r in rows[first - 3:last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - previous_end) / 1e3 if previous_end else 0.0
    print("%9.1f us  dur %8.1f us  gap %6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:70]))
    previous_end = e
