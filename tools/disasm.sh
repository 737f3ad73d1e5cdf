#!/bin/bash
# tools/disasm.sh <object under build/csrc> <kernel-name fragment>: gfx950 ISA of one kernel on stdout
set -e
obj=$(readlink -f "$1"); frag=$2
tmp=$(mktemp -d)
( cd "$tmp" && cp "$obj" o.o && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o > /dev/null 2>&1 || true )
dev=$(ls "$tmp"/o.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$dev" | awk -v f="$frag" '$0 ~ "<.*" f ".*>:" {on=1} on {print} on && /s_endpgm/ {exit}'
rm -rf "$tmp"
