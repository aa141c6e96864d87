"""Reads a rocprofv3 kernel trace (csv) and describes the steady state of a pipelined run: per kernel name the
count and mean duration, and over the last `window_ms` of the trace: how much of the time at least one kernel ran
(busy), the mean number of kernels running at once, the longest idle gaps, and the timeline of the last frames.

    python profiles/tools/timeline_overlap.py <trace dir> [window_ms] [rows]"""
import collections
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
window_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
show = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
shade = [i for i, r in enumerate(rows) if "shade_pixels" in r["Kernel_Name"]]
end_index = shade[-4] if len(shade) > 8 else len(rows) - 1   # stay clear of the drain at the end
t_end = int(rows[end_index]["Start_Timestamp"])
t_begin = t_end - int(window_ms * 1e6)
inside = [r for r in rows if int(r["End_Timestamp"]) > t_begin and int(r["Start_Timestamp"]) < t_end]
names = collections.defaultdict(list)
for r in inside:
    names[r["Kernel_Name"].split("(")[0][-60:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("window: last %.2f ms before the drain, %d kernels" % (window_ms, len(inside)))
for name, durations in sorted(names.items(), key=lambda kv: -sum(kv[1])):
    print("  %-62s n %4d  mean %8.1f us  sum %9.1f us" % (name, len(durations), sum(durations) / len(durations), sum(durations)))
events = []
for r in inside:
    events.append((max(int(r["Start_Timestamp"]), t_begin), 1))
    events.append((min(int(r["End_Timestamp"]), t_end), -1))
events.sort()
running, last, busy, weighted, gaps = 0, t_begin, 0, 0, []
for t, step in events:
    if running > 0:
        busy += t - last
        weighted += running * (t - last)
    elif t > last:
        gaps.append((t - last) / 1e3)
    running += step
    last = t
span = t_end - t_begin
frames = sum(1 for r in inside if "shade_pixels" in r["Kernel_Name"] and int(r["Start_Timestamp"]) >= t_begin)
print("busy %.1f %% of the window, mean kernels running while busy %.2f, idle gaps: %d, longest %s us" % (100.0 * busy / span, weighted / max(busy, 1), len(gaps), ", ".join("%.1f" % g for g in sorted(gaps)[-5:])))
print("shading launches started in the window: %d -> %.4f ms per launch" % (frames, window_ms / max(frames, 1)))
print("timeline of the last kernels (start relative to the first shown, duration, queue):")
tail = rows[max(0, end_index - show):end_index]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    print("  %9.1f us  dur %8.1f us  q %-4s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-50:]))
