"""Per hardware queue of a rocprofv3 kernel trace (csv): the kernels of the last 3 ms before the end, with the gap between
the end of a kernel and the start of the next one ON THE SAME QUEUE - where a stream sat idle between its own kernels."""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
end = max(int(r["End_Timestamp"]) for r in rows)
window = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows = [r for r in rows if int(r["Start_Timestamp"]) > end - window * 1e6 - 2e6 and int(r["End_Timestamp"]) < end - 2e6]
t0 = int(rows[0]["Start_Timestamp"])
last_end = {}
names = {"shade_pixels": "shade", "k_light_shafts": "shafts", "trace_shadow": "trace", "resolve_shadow": "resolve", "fillBuffer": "fill", "copyBuffer": "copy", "k_assemble": "scatter"}
stats = {}
for r in rows:
    q = r["Queue_Id"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = next((v for k, v in names.items() if k in r["Kernel_Name"]), r["Kernel_Name"][:24])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    print("%9.1f us  q %-3s %-8s dur %7.1f  gap on queue %7.1f" % ((s - t0) / 1e3, q, name, (e - s) / 1e3, gap))
    stats.setdefault(name, []).append(gap)
    last_end[q] = e
for name, gaps in stats.items():
    gaps = sorted(gaps)
    print("gap before %-8s n %3d  median %7.1f  mean %7.1f" % (name, len(gaps), gaps[len(gaps) // 2], sum(gaps) / len(gaps)))
