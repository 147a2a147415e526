#!/bin/bash
# SQ-level counters for the conv kernels (two passes of <= 8 SQ counters), denoiser forwards only.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/pmc_sq
rocprofv3 -L > $R/gpurun_out/pmc_sq/counters.txt 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/sq_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sq_$i -o sq -- python $R/tools/pmc_target.py 256 > $R/gpurun_out/pmc_sq/pass$i.log 2>&1
  echo "rc=$?" >> $R/gpurun_out/pmc_sq/pass$i.log
  f=$(find /tmp/sq_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_parse_multi.py $f > $R/gpurun_out/pmc_sq/pass$i.json
done
