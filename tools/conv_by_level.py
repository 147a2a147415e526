"""conv_f16ws launches of a rocprofv3 kernel trace bucketed by launch size (development aid; the configs[4] question: how much of
the convolution time sits in launches that do not fill the chip?).  The kernel is persistent: one workgroup per CU walks the
256-pixel tiles, so the grid is min(tiles, CUs) workgroups -- launches with fewer workgroups than CUs are the under-filled levels.
usage: python tools/conv_by_level.py kernel_trace.csv [csv_out]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
total = 0.0
for r in rows:
    name = r["Kernel_Name"]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    total += dur
    if "conv_f16ws_kernel" not in name and "attention_f16x2" not in name:
        continue
    wgs = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1))
    short = name.split("(")[0].replace("void ", "")
    # full launches: split by duration decade so that the 256^2 and 128^2 levels (both >= 256 workgroups) separate
    a = acc[(short, wgs)]
    a[0] += 1
    a[1] += dur
    a[2] = min(a[2], dur)
    a[3] = max(a[3], dur)
out = [("kernel", "workgroups", "launches", "total_ms", "avg_us", "min_us", "max_us", "share_of_all_kernel_time")]
for (k, w), (n, t, lo, hi) in sorted(acc.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    out.append((k, w, n, round(t / 1e3, 3), round(t / n, 1), round(lo, 1), round(hi, 1), round(t / total, 4)))
w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
w.writerows(out)
