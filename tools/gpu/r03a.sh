#!/bin/bash
# Round-3 session A: new conv_f16ws (asm-counted producer loads, half-tap consumer pipeline) vs the round-2 kernel.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03a; mkdir -p $O; cd $R
export TMPDIR=/tmp
L2=$R/diamond_amd/ablate/libdiamond_hip_r02.so
echo "=== conv tests (new)"; timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tpw.py tests/test_gpu_precision.py -m gpu -q -x -p no:cacheprovider > $O/tests_conv.log 2>&1; tail -3 $O/tests_conv.log
echo "=== all gpu tests (new)"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_all.log 2>&1; tail -3 $O/tests_all.log; grep -E "FAILED|ERROR" $O/tests_all.log | head -20
for v in r02 new r02 new; do lib=$R/diamond_amd/libdiamond_hip.so; [ $v = r02 ] && lib=$L2
  echo "=== conv_bench $v"; DIAMOND_LIB=$lib timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/conv_bench_$v.log; done
for n in 162 2 128; do echo "=== conv_bench ABL $n"; DIAMOND_LIB=$R/diamond_amd/ablate/libdiamond_hip_ws$n.so timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | head -4 | tee $O/conv_bench_abl$n.log; done
echo "=== conv_bench cout32"; for v in r02 new; do lib=$R/diamond_amd/libdiamond_hip.so; [ $v = r02 ] && lib=$L2; CONV_BENCH_COUT=32 DIAMOND_LIB=$lib timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_bench32_$v.log; done
for v in r02 new r02 new; do lib=$R/diamond_amd/libdiamond_hip.so; [ $v = r02 ] && lib=$L2
 DIAMOND_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench_$v.json 2> $O/bench_$v.err; python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v bench', d['value'], d['roofline']['avg_launch_ms'])"; done
echo "=== SQ counters"
bash tools/pmc_sq.sh new $R/diamond_amd/libdiamond_hip.so
bash tools/pmc_sq.sh abl162 $R/diamond_amd/ablate/libdiamond_hip_ws162.so
bash tools/pmc_sq.sh r02 $L2
for t in new abl162 r02; do echo "-- $t"; python - <<PY
import json
for i in (1,2):
    try:
        d=json.load(open("$R/gpurun_out/pmc_sq_$t/pass%d.json"%i))
        for k,v in d.items():
            if "WsGeom<false, 2, 9" in k: print(i, {a:round(b) for a,b in v.items()})
        print(json.load(open("$R/gpurun_out/pmc_sq_$t/pass%d_durations.json"%i)))
    except Exception as e: print("ERR", e)
PY
done
echo "=== config 4 profile"
(cd /tmp && rm -rf /tmp/prof_cfg4 && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg4 -o cfg4 -- python $R/bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/cfg4_prof.log 2>&1; echo "rocprof rc=$?"; f=$(find /tmp/prof_cfg4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/cfg4_kernel_stats.csv && head -16 $f)
tail -2 $O/cfg4_prof.log | cut -c1-400
