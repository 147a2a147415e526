#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
A=$R/diamond_amd/ablate
DIAMOND_LIB=$A/libdiamond_hip_r02.so python tools/debug/r03_ac_debug.py run /tmp/ac_r02.pt 2>&1 | grep -v amdgpu.ids
DIAMOND_LIB=$R/diamond_amd/libdiamond_hip.so python tools/debug/r03_ac_debug.py run /tmp/ac_new.pt 2>&1 | grep -v amdgpu.ids
python tools/debug/r03_ac_debug.py cmp /tmp/ac_new.pt /tmp/ac_r02.pt 2>&1 | tee $O/ac_cmp.log
