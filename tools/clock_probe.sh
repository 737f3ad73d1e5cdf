#!/bin/bash
# usage (GPU box, repo root): LAYERS=44 HINTS=12,14:196 bash tools/clock_probe.sh <outdir>
set -e
OUT=${1:-gpurun_out/clock}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT" -o p -- python3 tools/time_igemm.py > "$OUT/run.txt" 2> "$OUT/run.err"
python3 tools/clock_probe.py "$OUT"/p_counter_collection.csv
