#!/bin/bash
# detailed PMC passes over tools/kbench.py for the backward kernels: tools/pmc_detail.sh TAG [env assignments...]
TAG=$1; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" \
           "TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  env "$@" rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/${TAG}_p$i -- python tools/kbench.py 8192 512 bf16 > gpurun_out/${TAG}_p$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/${TAG}_p*/*/*_counter_collection.csv > gpurun_out/${TAG}_pmc_summary.txt 2>&1
grep "bwd\|fwd_pipe" gpurun_out/${TAG}_pmc_summary.txt
