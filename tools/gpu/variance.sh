#!/bin/bash
# Run-to-run spread of the headline number on ONE box: the same command N times (what an A/B delta has to be read against).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/variance; mkdir -p $O; : > $O/runs.txt
for i in $(seq 1 ${N:-8}); do
  timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-also --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'], 1), round(d['ms_per_step'], 2))" | tee -a $O/runs.txt
done
python - <<'PY'
import statistics as st
v = [float(l.split()[0]) for l in open("gpurun_out/variance/runs.txt") if l.strip()]
print(f"n={len(v)} mean {st.mean(v):.1f} stdev {st.stdev(v):.1f} ({100*st.stdev(v)/st.mean(v):.2f} %) min {min(v):.1f} max {max(v):.1f}; without the first run: mean {st.mean(v[1:]):.1f} stdev {st.stdev(v[1:]):.1f}")
PY
