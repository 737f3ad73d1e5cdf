#!/bin/bash
# SQ / LDS counters of the wgrad kernels on one layer shape.  usage (GPU box, repo root): bash tools/pmc_wgrad.sh <outdir> <layer idx> <variant>
set -e
OUT=${1:-gpurun_out/pmc_wgrad}; export LAYERS=${2:-12}; export VARIANT=${3:-1}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d "$OUT" -o p1 -- python3 tools/time_wgrad.py > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM TA_BUSY_avr --output-format csv -d "$OUT" -o p2 -- python3 tools/time_wgrad.py > "$OUT/p2.log" 2>&1
ls "$OUT"
