#!/bin/sh
# Builds libazgomoku.so for gfx950 (MI355X).
#  -ffp-contract=off keeps the PUCT and running-mean arithmetic in the reference's operation
#   order (no fused multiply-add).
#  -no-hip-rt leaves the HIP runtime symbols undefined: the process must hold exactly ONE HIP
#   runtime, and PyTorch ships its own libamdhip64.so.  The loader (alphazero_gomoku_amd/_lib.py)
#   loads the runtime PyTorch will use (or /opt/rocm's when PyTorch is absent) with RTLD_GLOBAL
#   before this library; a C/C++ host simply links -lamdhip64 itself.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../libazgomoku.so"
log="$(mktemp)"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wall -no-hip-rt \
    -Rpass-analysis=kernel-resource-usage \
    "$here/azgomoku.hip" -o "$out" -lpthread "$@" 2> "$log" || { cat "$log" >&2; rm -f "$log"; exit 1; }
# compiler warnings, without the resource-usage remarks and their source excerpts
if grep -q "warning:" "$log"; then
    awk '/kernel-resource-usage/ { skip = 2; next } skip > 0 { skip--; next } { print }' "$log" >&2
fi
# The split-f16 search kernels for boards up to 8x8 run two workgroups per CU.  Up to 240 VGPRs,
# two waves per SIMD leave 32 of the 512 registers free, which a wave of the runtime's copy kernel
# needs to run BESIDE the search (the noise upload of the pipelined move step); at 248 the copy
# waits for the whole search launch and a move step grows by 0.3 ms (measured).
awk '/Function Name: _ZN3azg13search_kernelILi[68]ELi2ELi0ELb1EEEvNS_8DevStateE/ { k = $0; sub(/.*Function Name: /, "", k); sub(/ .*/, "", k) }
     k != "" && / VGPRs: / { v = $0; sub(/.* VGPRs: /, "", v); sub(/ .*/, "", v);
         if (v + 0 > 240) printf("WARNING: %s uses %d VGPRs (> 240): copy kernels cannot run beside it\n", k, v);
         k = "" }' "$log" >&2
rm -f "$log"
echo "built $out"
