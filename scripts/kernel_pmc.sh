#!/bin/bash
# PMC study of one kernel class: what the request path, the LDS and the matrix pipe do while it runs.  Run ON THE GPU BOX from
# the repo root:   bash scripts/kernel_pmc.sh <tag> <kernel-name substring> python3 <program> [args]
# e.g.             bash scripts/kernel_pmc.sh r04_x3 "conv_mfma_kernel<8" python3 scripts/conv_single.py f32x3 10
# Counters go in separate passes, never with a trace domain other than --kernel-trace (MI355X_MICROARCH.md, rocprofv3 PMC
# slots; gpurun rules); the program after `--` is python3 itself.  Summaries land in gpurun_out/pmc_<tag>/.
set -e
TAG=$1; PAT=$2; shift 2
OUT=gpurun_out/pmc_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for ctr in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctr --output-format rocpd -d $OUT/p$i -o p$i -- "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i ($ctr) failed"; tail -3 $OUT/p$i.log; continue; }
  python3 scripts/rocpd_pmc_summary.py $(find $OUT/p$i -name "*.db" | head -1) $OUT/pmc_p$i.md > /dev/null
  rm -rf $OUT/p$i
  echo "pass $i done: $ctr"
done
cat $OUT/pmc_p*.md | grep -E "counter|dispatches|$PAT" | cut -c1-260
