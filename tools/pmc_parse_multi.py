"""Aggregate a rocprofv3 counter_collection.csv with several counters: kernel -> counter -> mean per launch."""
import csv, json, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    a = acc[r["Kernel_Name"]][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
out = {k: {c: v[1] / v[0] for c, v in d.items()} | {"launches": max(v[0] for v in d.values())} for k, d in acc.items()
       if "conv_" in k or "wgrad" in k or "attention" in k or "linear" in k}
print(json.dumps(out, indent=1))
