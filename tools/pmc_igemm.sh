#!/bin/bash
# SQ / LDS counters of the forward's kernels (bench.py --no-train --no-nms): two separate --pmc passes, never combined with tracing.
# usage (GPU box, repo root): bash tools/pmc_igemm.sh <outdir>
set -e
OUT=${1:-gpurun_out/pmc_igemm}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT" -o p1 -- python3 bench.py --steps 3 --warmup 1 --no-train --no-nms --no-cpu --no-resnet > "$OUT/p1.json" 2> "$OUT/p1.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT" -o p2 -- python3 bench.py --steps 3 --warmup 1 --no-train --no-nms --no-cpu --no-resnet > "$OUT/p2.json" 2> "$OUT/p2.err"
python3 tools/pmc_table.py "$OUT"/p1_counter_collection.csv "$OUT"/p2_counter_collection.csv
