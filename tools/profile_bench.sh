#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 passes over the default bench.py invocation.
#   pass 1: --kernel-trace --stats                (per-kernel durations)
#   pass 2: --pmc FETCH_SIZE  (+ kernel trace)    (HBM read traffic; own pass: 3 TCC slots)
#   pass 3: --pmc WRITE_SIZE  (+ kernel trace)    (HBM write traffic)
#   pass 4/5: --pmc SQ_* issue / wait / LDS counters
#   pass 6: --pmc FP64 instruction mix + GRBM_GUI_ACTIVE (clock)
# PMC passes never carry --stats or any trace domain besides --kernel-trace (the pool refuses that).
# Output under gpurun_out/<tag>/ ; tools/summarize_profile.py turns it into profiles/<tag>_*.
set -u
TAG=${1:-prof}
shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-mintime --no-qss $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats -- python3 "$R/bench.py" $ARGS > "$O/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O" -o fetch -- python3 "$R/bench.py" $ARGS > "$O/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O" -o write -- python3 "$R/bench.py" $ARGS > "$O/write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d "$O" -o sq1 -- python3 "$R/bench.py" $ARGS > "$O/sq1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --kernel-trace --output-format csv -d "$O" -o sq2 -- python3 "$R/bench.py" $ARGS > "$O/sq2.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O" -o sq3 -- python3 "$R/bench.py" $ARGS > "$O/sq3.log" 2>&1
ls "$O"
