"""Per-step timeline of a rocprofv3 --kernel-trace CSV: for the LAST complete optimizer step (from one adam_kernel to
the next) print every kernel with start offset, duration and stream/queue, plus the sum of durations, the wall time
of the step and the idle gaps -- shows what runs next to what and where the step waits.
usage: trace_timeline.py kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
if len(adam) < 3:
    sys.exit("fewer than three optimizer steps in the trace")
a, b = adam[-3], adam[-2]
step = rows[a + 1:b + 1]
t0 = step[0]["s"]
busy = 0
last_end = t0
gaps = 0
print("%9s %9s %6s  %s" % ("start_us", "dur_us", "queue", "kernel"))
for r in step:
    name = r["Kernel_Name"]
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    print("%9.1f %9.1f %6s  %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r.get("Queue_Id", "?"), name[:110]))
    busy += r["e"] - r["s"]
    if r["s"] > last_end:
        gaps += r["s"] - last_end
    last_end = max(last_end, r["e"])
print("kernels %d  sum of durations %.1f us  wall %.1f us  idle gaps %.1f us" % (len(step), busy / 1e3, (last_end - t0) / 1e3, gaps / 1e3))
