#!/bin/bash
# SQ-level counters for the conv kernels (passes of <= 8 counters of one block each: SQ x 4, TCC, TCP / TA), denoiser forwards only.
#   bash tools/pmc_sq.sh <tag> [lib.so]     -> gpurun_out/pmc_sq_<tag>/pass{1,2}.json
#   PMC_TARGET="tools/wgrad_bench.py wgrad 3 f2" PMC_KERNELS=wgrad bash tools/pmc_sq.sh <tag>     (another workload / kernel family)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-head}
[ -n "${2:-}" ] && export DIAMOND_LIB=$2
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/pmc_sq_$TAG
mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LEVEL_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  rm -rf /tmp/sq_${TAG}_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sq_${TAG}_$i -o sq -- python $R/${PMC_TARGET:-tools/pmc_target.py 256} > $O/pass$i.log 2>&1
  echo "rc=$?" >> $O/pass$i.log
  f=$(find /tmp/sq_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_parse_multi.py $f > $O/pass$i.json
  k=$(find /tmp/sq_${TAG}_$i -name "*kernel_trace.csv" | head -1)
  [ -n "$k" ] && python - "$k" "${PMC_KERNELS:-conv}" > $O/pass${i}_durations.json <<'PY'
import csv, json, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(json.dumps({k: {"launches": len(v), "avg_us": sum(v) / len(v) / 1e3} for k, v in d.items() if sys.argv[2] in k}, indent=1))
PY
done
