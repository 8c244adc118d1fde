#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 counter passes over tools/time_qss.py with both simulator kernels forced in turn
# (256 Monza tables of N = 2000 / N = 500 and one alone).  PMC passes carry --kernel-trace only.  Output: gpurun_out/<tag>/.
set -u
TAG=${1:-qssprof}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d "$O" -o sq1 -- python3 "$R/tools/time_qss.py" 1 256 > "$O/sq1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --kernel-trace --output-format csv -d "$O" -o sq2 -- python3 "$R/tools/time_qss.py" 1 256 > "$O/sq2.log" 2>&1
ls "$O"
