"""Per-launch durations and start-to-start intervals of the rollout kernel by launch index, from a rocprofv3 kernel trace
(rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 2000 --warmup 5 ...): does the kernel run at one
speed from the first launch on?   usage: python tools/launch_timeline.py DIR/t_kernel_trace.csv"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_rollout" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
print("launches", len(rows))
edges = [0, 5, 25, 50, 100, 200, 400, 800, 1200, 1600, 2000, 2400]
for a, b in zip(edges, edges[1:]):
    b = min(b, len(rows))
    if b - a < 2:
        break
    dur = sum(en[i] - st[i] for i in range(a, b)) / (b - a) / 1e3
    gap = (st[b - 1] - st[a]) / (b - 1 - a) / 1e3
    print("launch %4d..%4d  duration %.2f us  start-to-start %.2f us  (t = %.2f ms)" % (a, b, dur, gap, (st[a] - st[0]) / 1e6))
