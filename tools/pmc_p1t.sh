#!/bin/bash
# SQ / GRBM counters of the pointwise backward-weight kernel:  tools/pmc_p1t.sh <tag> <rows,K,N>   (run on the GPU box)
set -u
TAG=$1; SHAPE=$2
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d "$O/p1" -- python3 "$R/tools/run_p1t.py" 16 $SHAPE > "$O/p1.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/p2" -- python3 "$R/tools/run_p1t.py" 16 $SHAPE > "$O/p2.log" 2>&1
python3 "$R/tools/pmc_summary.py" p1t_kernel "$O/p1" "$O/p2" > "$O/summary.json"
cat "$O/summary.json"
rm -rf "$O"/p? 2>/dev/null
