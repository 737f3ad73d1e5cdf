#!/bin/bash
# One profile set of a round, on the GPU box from the repo root:  bash tools/profile_round.sh r2_v1
#   1. rocprofv3 --kernel-trace --stats of the default bench.py line  -> profiles/<tag>_bench_kernel_stats.{csv,txt}, <tag>_bench.json,
#      <tag>_roofline_pass_from_trace.txt (kernel-trace durations of the roofline pass vs bench.py's own HIP events)
#   2. HBM traffic of the forward's igemm launches (two separate --pmc passes) -> profiles/<tag>_igemm_traffic.json
#   3. SQ counters (MFMA busy, LDS conflicts, issue stalls; two --pmc passes)   -> profiles/<tag>_igemm_sq_pmc.txt
#   4. the same for the weight-gradient kernels over three training steps       -> profiles/<tag>_wgrad_traffic.json, <tag>_wgrad_pmc.txt
#   5. kernel stats of the training step alone                                  -> profiles/<tag>_train_step_kernels.txt
#   6. kernel stats of 200 single-image forwards                                -> profiles/<tag>_batch1_kernels.txt
# Counter passes never run together with tracing (MI355X_MICROARCH.md, rocprofv3 PMC).
set -e
TAG=${1:-r2_v1}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT" profiles
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- python3 bench.py --no-cpu > "$OUT/bench.json" 2> "$OUT/bench.err"
cp "$OUT/bench.json" profiles/${TAG}_bench.json
cp "$OUT"/bench_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
python3 tools/kstats.py "$OUT"/bench_kernel_stats.csv 45 > profiles/${TAG}_bench_kernel_stats.txt
python3 tools/trace_roofline.py "$OUT"/bench_kernel_trace.csv > profiles/${TAG}_roofline_pass_from_trace.txt
cat profiles/${TAG}_roofline_pass_from_trace.txt
bash tools/collect_traffic.sh "$OUT/traffic" > /dev/null
cp "$OUT/traffic/igemm_traffic.json" profiles/${TAG}_igemm_traffic.json
cat profiles/${TAG}_igemm_traffic.json
bash tools/pmc_igemm.sh "$OUT/pmc" > profiles/${TAG}_igemm_sq_pmc.txt
tail -20 profiles/${TAG}_igemm_sq_pmc.txt
bash tools/collect_wgrad_traffic.sh "$OUT/wgrad" > /dev/null
cp "$OUT/wgrad/wgrad_traffic.json" profiles/${TAG}_wgrad_traffic.json
cp "$OUT/wgrad/wgrad_pmc.txt" profiles/${TAG}_wgrad_pmc.txt
cat profiles/${TAG}_wgrad_traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o train -- python3 tools/train_steps.py 10 > "$OUT/train.log" 2>&1
python3 tools/kstats.py "$OUT"/train_kernel_stats.csv 40 > profiles/${TAG}_train_step_kernels.txt
python3 bench.py > profiles/${TAG}_bench_unprofiled.json 2> /dev/null
# 6. the single-image forward (predict.py; BASELINE configs[0] runs it on the CPU): kernel stats of 200 batch-1 forwards
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o b1 -- python3 tools/small_batch_steps.py 200 1 > "$OUT/b1.log" 2>&1
{ cat "$OUT/b1.log" | grep "ms per forward"; python3 tools/kstats.py "$OUT"/b1_kernel_stats.csv 30; } > profiles/${TAG}_batch1_kernels.txt
