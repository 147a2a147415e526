#!/bin/bash
# Kernel-only durations of the weight-gradient launches of one shape (rocprofv3 --stats of tools/wgrad_bench.py).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-wgrad_prof}; mkdir -p $O
export TMPDIR=/tmp
i=0
for shape in "f2 64^2 64->64" "f2 64^2 64->64 raw" "f2 16^2 64->64" "ac 64^2 32->32" "ac 8^2 64->64"; do
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/wp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wp -o t -- python $R/tools/wgrad_bench.py wgrad 20 "$shape" > $O/s$i.log 2>&1)
  k=$(find /tmp/wp -name "*kernel_stats.csv" | head -1)
  echo "== $shape"; grep wgrad $O/s$i.log | cut -c1-80
  python - "$k" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "wgrad" in r["Name"]:
        print(f"   {int(r['Calls']):4d}x avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f}  {r['Name'][:90]}")
P
done 2>&1 | tee $O/summary.txt
