#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 passes over the batched min-time solve (tools/bench_mintime.py B).
#   pass 1: --kernel-trace --stats                         (per-kernel durations)
#   pass 2: --pmc matrix-core / issue counters             (+ kernel trace only: the pool refuses other domains)
#   pass 3: --pmc FP64 instruction mix                     (executed FP64 flops: the roofline of bench.py's mintime_nlp leg)
# Output under gpurun_out/<tag>/ ; the per-kernel summary is written to profiles/<tag>_kernels.json by
# tools/summarize_mintime_profile.py.
set -u
TAG=${1:-mt}
B=${2:-256}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats -- python3 "$R/tools/bench_mintime.py" $B > "$O/stats.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d "$O" -o pmc -- python3 "$R/tools/bench_mintime.py" $B > "$O/pmc.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O" -o fp64 -- python3 "$R/tools/bench_mintime.py" $B > "$O/fp64.log" 2>&1
ls "$O"
