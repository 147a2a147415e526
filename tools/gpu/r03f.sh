#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
export TMPDIR=/tmp
echo "=== attention tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "attention or attn or 256" > $O/tests_att.log 2>&1; tail -3 $O/tests_att.log; grep -E "FAILED|ERROR" $O/tests_att.log | head
for v in 0 1; do echo "=== config 4 bench, DIAMOND_ATTENTION_F16X2=$v"; DIAMOND_ATTENTION_F16X2=$v timeout 300 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2> $O/cfg4_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
(cd /tmp && rm -rf /tmp/prof_cfg4 && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg4 -o cfg4 -- python $R/bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/cfg4_prof.log 2>&1; echo "rocprof rc=$?"; f=$(find /tmp/prof_cfg4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/cfg4_kernel_stats.csv && head -12 $f | cut -c1-160)
