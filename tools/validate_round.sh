#!/bin/bash
# end-of-round check on the GPU box: smoke(), the default bench line (wall time), the whole GPU suite
S=$(date +%s)
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$? wall=$(( $(date +%s) - S ))s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["train"]["ms_per_step"], d["cpu_baseline"], d["roofline"]["frac"], (d.get("resnet50_variant") or {}).get("value"))
PY
timeout -k 10 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
