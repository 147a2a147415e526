#!/bin/bash
# gpurun_out/<tag>/ (tools/gpu/r06_final.sh <tag>) -> profiles/<tag>_*: the files profiles/README.md lists for the shipped library.
#   bash tools/evidence_to_profiles.sh r06o
set -eu
cd "$(dirname "$0")/.."
TAG=$1; S=gpurun_out/$TAG
for f in tests.log smoke.log box.txt bench_driver_line.json bench_kernel_stats.csv steady_p003_gaps.txt no_ends_gaps.txt; do
  cp $S/$f profiles/${TAG}_$f
done
mkdir -p gpurun_out/pmc gpurun_out/pmc_cfg4
cp $S/pmc/*.json gpurun_out/pmc/; cp $S/pmc_cfg4/*.json gpurun_out/pmc_cfg4/
python tools/pmc_to_profile.py $TAG > /dev/null
python tools/pmc_to_profile.py $TAG 4 > /dev/null
python - $TAG <<'PY'
import json, sys
d = json.loads(open(f"profiles/{sys.argv[1]}_bench_driver_line.json").read().strip().splitlines()[-1])
r, a = d["roofline"], d["also"]
print("value", round(d["value"], 1), "ms_per_step", round(d["ms_per_step"], 2), "| dominant", r["kernel"], round(1e3 * r["avg_launch_ms"], 2), "us frac", round(r["frac"], 4), "traffic", r.get("traffic"))
print("no_ends", round(a["end_rate"]["no_ends"]["value"], 1), "| exact_fp32", d.get("exact_fp32", {}).get("value"), "| cfg3", round(a["configs[3]"]["value"], 1), "cfg4", round(a["configs[4]"]["value"], 1),
      "| latency", round(a["latency"]["value"], 3), "| train", round(a["train"]["value"], 3), "| cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY
