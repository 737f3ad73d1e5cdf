#!/bin/bash
# VGPRs / scratch bytes / occupancy of every kernel of one source file (cross-compiled, no GPU needed):
#   bash tools/kernel_resources.sh igemm.hip
# Run it after every change to a kernel: at 246-256 VGPRs a few extra epilogue lines make hipcc spill around the K loop or
# take a thin-K configuration from five co-resident workgroups per CU to two (round 2: ResNet-50 inference 5.99 -> 7.29 ms).
set -e
cd "$(dirname "$0")/../yolo-v1_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -c "$1" -o /tmp/kernel_resources.o -Rpass-analysis=kernel-resource-usage 2>&1 |
    grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | paste - - - - |
    sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/[a-z_]*\.hip:[0-9]*:[0-9]*: remark://g' |
    awk '{printf "%-110s vgpr %4s  scratch %4s  waves/SIMD %s\n", $3, $5, $8, $12}' | sort
