#!/bin/bash
# PMC study of conv1x1_h256_kernel (the dominant kernel of BASELINE configs[4]): what the L2 -> LDS request path and the
# matrix pipe do during the kernel.  Run ON THE GPU BOX from the repo root:   bash scripts/h256_pmc.sh r04
# Counters go in separate passes, never with a trace domain other than --kernel-trace (MI355X_MICROARCH.md, rocprofv3 PMC
# slots; gpurun rules); the program after `--` is python3 itself.  Summaries land in gpurun_out/h256_pmc_<tag>/.
set -e
TAG=${1:-r04}
OUT=gpurun_out/h256_pmc_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for ctr in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctr --output-format rocpd -d $OUT/p$i -o p$i -- python3 scripts/h256_single.py 10 > $OUT/p$i.log 2>&1 || { echo "pass $i ($ctr) failed"; tail -3 $OUT/p$i.log; continue; }
  python3 scripts/rocpd_pmc_summary.py $(find $OUT/p$i -name "*.db" | head -1) $OUT/pmc_p$i.md > /dev/null
  rm -rf $OUT/p$i
  echo "pass $i done: $ctr"
done
cat $OUT/pmc_p*.md | grep -E "counter|h256|^\| k" | cut -c1-250
