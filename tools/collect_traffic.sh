#!/bin/bash
# HBM traffic of the dominant kernel (igemm) for bench.py's forward, from PMC counters.
# Two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with
# tracing), as MI355X_MICROARCH.md "HBM" prescribes.  Run on the GPU box from the repo root:
#     bash tools/collect_traffic.sh gpurun_out/traffic
set -e
OUT=${1:-gpurun_out/traffic}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT" -o pmc_$c -- python3 bench.py --steps 3 --warmup 1 --no-train --no-nms --no-cpu --no-resnet > "$OUT/bench_$c.json" 2> "$OUT/err_$c.log"
done
python3 tools/traffic_summary.py "$OUT" > "$OUT/igemm_traffic.json"
cat "$OUT/igemm_traffic.json"
