#!/bin/bash
# Which clock does the chip run under which instruction mix?  Four readings side by side (profiles/r04_clock.json):
# throughput-derived (tools/probe/clock_probe: a known instruction count / time), s_memtime ticks / event time, the SMI /
# sysfs sensors sampled while the kernel runs (tools/smi_sampler.py), and -- from the SQ / GRBM counters of
# profiles/r03_sq_counters.json -- the counter-derived ones.  ~3 GPU-minutes.
#   gpurun --timeout 600 -- 'bash tools/gpu/clock_table.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/clock; mkdir -p $O
sampled() {  # label, command...
  local label=$1; shift
  python tools/smi_sampler.py $O/$label.csv 20 > $O/$label.smi.json 2>/dev/null &
  local sp=$!
  sleep 0.5
  timeout 120 "$@" > $O/$label.log 2>&1
  kill $sp; wait $sp 2>/dev/null
  echo "== $label"; tail -3 $O/$label.log; cat $O/$label.smi.json; echo
}
(rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null; amd-smi static --limit 2>/dev/null | head -60) > $O/idle.txt 2>&1
sampled mfma_random tools/probe/clock_probe mfma 6 1
sampled mfma_zero tools/probe/clock_probe mfma 6 0
sampled valu_random tools/probe/clock_probe valu 4 1
sampled conv64 python tools/conv_loop.py 8 0 64 64 0
sampled conv64_zero python tools/conv_loop.py 8 1 64 64 0
sampled conv32 python tools/conv_loop.py 6 0 32 32 1
sampled conv128 python tools/conv_loop.py 6 0 128 64 0
# s_memtime timelines of the instances that have none yet (and the ticks / us of each)
if [ -f diamond_amd/ablate/libdiamond_hip_wstrace.so ]; then
  for args in "64 0 64 64" "64 1 64 64" "128 0 64 64" "64 2 64 64" "32 1 32 64" "32 0 32 64" "64 0 64 32" "64 0 64 16"; do
    echo "== ws_trace $args"
    DIAMOND_LIB=diamond_amd/ablate/libdiamond_hip_wstrace.so timeout 120 python tools/ws_trace.py $args 2>&1 | tail -4 | tee $O/ws_trace_$(echo $args | tr ' ' '_').txt
  done
fi
