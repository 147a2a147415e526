"""Per-kernel launch durations of a rocprofv3 kernel trace, clustered by duration (development aid): a training step launches the
same kernel on every U-Net level, and the average over levels says nothing about the 64x64 launches that carry the time.
usage: python tools/trace_by_shape.py kernel_trace.csv [name substring ...]   (last LAST_MS=30 ms of the trace)"""
import csv, os, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:] or ["wgrad_kernel", "gn_bwd", "conv_f16ws", "conv1x1", "conv_mfma", "wgrad_reduce"]
end = max(int(r["End_Timestamp"]) for r in rows)
last = float(os.environ.get("LAST_MS", "30")) * 1e6
by = defaultdict(list)
for r in rows:
    if int(r["Start_Timestamp"]) < end - last:
        continue
    n = r["Kernel_Name"]
    if any(w in n for w in want):
        by[(n[:70], r.get("Grid_Size", r.get("Grid_Size_X", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in by.values())
for (n, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{sum(v)/1e3:8.3f} ms {len(v):5d}x grid {g:>9}  min {v[0]:7.1f} med {v[len(v)//2]:7.1f} max {v[-1]:7.1f}  {n}")
    if len(v) <= 80:
        print("          " + " ".join(f"{x:.0f}" for x in v))
