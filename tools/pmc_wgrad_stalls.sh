#!/bin/bash
# stall counters (LDS / VMEM queues, issue) of the wgrad kernels on one layer shape.  usage (GPU box, repo root): bash tools/pmc_wgrad_stalls.sh <outdir> <layer idx> <variant>
set -e
OUT=${1:-gpurun_out/pmc_wgrad_stalls}; export LAYERS=${2:-12}; export VARIANT=${3:-6}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT" -o p1 -- python3 tools/time_wgrad.py > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d "$OUT" -o p2 -- python3 tools/time_wgrad.py > "$OUT/p2.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d "$OUT" -o p3 -- python3 tools/time_wgrad.py > "$OUT/p3.log" 2>&1
ls "$OUT"
