#!/bin/bash
# kernel mix of the ResNet-50 variant's training step (the reference's default training model): rocprofv3 --kernel-trace --stats of 6 steps
#     bash tools/resnet_train_kernels.sh gpurun_out/resnet_train      (GPU box, repo root)
set -e
OUT=${1:-gpurun_out/resnet_train}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MODEL=resnet50
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o rt -- python3 tools/train_steps.py 6 > "$OUT/steps.log" 2>&1
python3 tools/kstats.py "$OUT"/rt_kernel_stats.csv 40 > "$OUT/resnet_train_kernels.txt"
cat "$OUT/resnet_train_kernels.txt"
