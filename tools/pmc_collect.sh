#!/bin/bash
# HBM traffic counters for the hot kernels, one counter per pass (FETCH_SIZE uses 3 of the 4 TCC
# slots, WRITE_SIZE 2: they cannot share a pass; no trace domains besides --kernel-trace).
# The profiled workload is ONE BENCH WINDOW of configs[1] (bench.py --steps 1 --warmup 1, preceded by a calibration copy of
# known size), i.e. the launches bench.py's roofline averages over -- not a stand-in.
# Run on the GPU box:  bash tools/pmc_collect.sh   -> gpurun_out/pmc/{FETCH,WRITE}_SIZE.json
#                      CFG=4 bash tools/pmc_collect.sh -> gpurun_out/pmc_cfg4/... (one window of configs[4]: 256x256, attention)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
CFG=${CFG:-1}
O=$R/gpurun_out/pmc; [ "$CFG" != 1 ] && O=$R/gpurun_out/pmc_cfg$CFG
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/bench.py --config $CFG --steps 1 --warmup 1 --pmc-calibrate --no-cpu-baseline --no-exact-fp32 --no-roofline --no-also > $O/$c.log 2>&1
  echo "rc=$?" >> $O/$c.log
  ls -la /tmp/pmc_$c >> $O/$c.log
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_parse.py $f $c > $O/$c.json
done
