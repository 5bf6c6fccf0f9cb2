#!/bin/bash
# SQ counters of the fused classifier head's three kernels:  tools/pmc_headfuse.sh <tag>   (run on the GPU box)
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d "$O/p1" -- python3 "$R/tools/run_headfuse.py" 8 > "$O/p1.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/p2" -- python3 "$R/tools/run_headfuse.py" 8 > "$O/p2.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d "$O/p3" -- python3 "$R/tools/run_headfuse.py" 8 > "$O/p3.log" 2>&1
for k in "hf_fwd_kernel" "hf_bwd_kernel<false" "hf_bwd_kernel<true"; do
  n=$(echo $k | tr -d '<' )
  python3 "$R/tools/pmc_summary.py" "$k" "$O/p1" "$O/p2" "$O/p3" > "$O/summary_$n.json"
  cat "$O/summary_$n.json"
done
rm -rf "$O"/p? 2>/dev/null
