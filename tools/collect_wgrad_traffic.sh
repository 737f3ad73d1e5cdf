#!/bin/bash
# HBM traffic and SQ counters of the weight-gradient kernels over three batch-64 training steps (tools/train_steps.py).
# Separate rocprofv3 --pmc passes, never combined with tracing (MI355X_MICROARCH.md "HBM", "rocprofv3 PMC slots").
#     bash tools/collect_wgrad_traffic.sh gpurun_out/wgrad_traffic    (GPU box, repo root)
set -e
OUT=${1:-gpurun_out/wgrad_traffic}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT" -o pmc_$c -- python3 tools/train_steps.py 3 > "$OUT/steps_$c.log" 2> "$OUT/err_$c.log"
done
python3 tools/traffic_summary.py "$OUT" wgrad > "$OUT/wgrad_traffic.json"
cat "$OUT/wgrad_traffic.json"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT" -o sq1 -- python3 tools/train_steps.py 3 > "$OUT/sq1.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d "$OUT" -o sq2 -- python3 tools/train_steps.py 3 > "$OUT/sq2.log" 2>&1
python3 tools/pmc_table.py "$OUT"/sq1_counter_collection.csv "$OUT"/sq2_counter_collection.csv > "$OUT/wgrad_pmc.txt"
