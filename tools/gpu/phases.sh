#!/bin/bash
# Host-issue vs wall time of a window's phases (tools/debug/window_phases.py) and the idle-gap traces (tools/gpu/gaps.sh)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $R
O=$R/gpurun_out/${1:-phases}; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/debug/window_phases.py > $O/window_phases.txt 2>&1; cat $O/window_phases.txt | tail -6
bash tools/gpu/gaps.sh ${1:-phases}/gaps
