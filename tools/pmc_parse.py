"""Aggregate a rocprofv3 counter_collection.csv: per kernel name, launches and mean counter value."""
import csv, json, sys
from collections import defaultdict
path, counter = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(path)):
    if r.get("Counter_Name") != counter:
        continue
    a = acc[r["Kernel_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
print(json.dumps({k: {"launches": v[0], "mean": v[1] / v[0], "total": v[1]} for k, v in acc.items()}, indent=1))
