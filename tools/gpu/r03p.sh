#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r03p; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/tr; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o tr -- python $R/bench.py --config train --steps 10 > $O/train_prof.log 2>&1
f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/train_kernel_stats.csv && head -30 $f | cut -c1-200
rm -rf /tmp/lt; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -o lt -- python $R/bench.py --config latency --steps 50 > $O/lat_prof.log 2>&1
f=$(find /tmp/lt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/lat_kernel_stats.csv && head -30 $f | cut -c1-200
tail -2 $O/train_prof.log | cut -c1-400; tail -2 $O/lat_prof.log | cut -c1-400
