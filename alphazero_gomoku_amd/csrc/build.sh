#!/bin/sh
# Builds libazgomoku.so for gfx950 (MI355X).
#  -ffp-contract=off keeps the PUCT and running-mean arithmetic in the reference's operation
#   order (no fused multiply-add).
#  -fno-slp-vectorize keeps float32 arithmetic on single-lane VALU instructions.  Left on, -O3 packs
#   adjacent multiply-adds into v_pk_fma_f32; in the kernel whose tree waves share their SIMDs with
#   matrix-dense waves such an operation, fed its x from the high dword of a register pair
#   (op_sel:[0,1,0]), came out one term short in lanes 48-55 (50-70 of 256 identical searches wrong;
#   the note at AZG_EXP_PKFC in azg_dual.hpp has the measurements).  The build FAILS when a search or
#   evaluation kernel holds any two-wide float32 instruction (v_pk_fma_f32 / v_pk_mul_f32 /
#   v_pk_add_f32), whatever flag or compiler version produced it.
#  -no-hip-rt leaves the HIP runtime symbols undefined: the process must hold exactly ONE HIP
#   runtime, and PyTorch ships its own libamdhip64.so.  The loader (alphazero_gomoku_amd/_lib.py)
#   loads the runtime PyTorch will use (or /opt/rocm's when PyTorch is absent) with RTLD_GLOBAL
#   before this library; a C/C++ host simply links -lamdhip64 itself.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../libazgomoku.so"
log="$(mktemp)"
tmp="$(mktemp -d)"
trap 'rm -rf "$tmp"' EXIT
# (the device assembly is kept as a by-product of the one compilation: -save-temps, in a scratch directory)
( cd "$tmp" && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 -Wall -no-hip-rt \
    -Rpass-analysis=kernel-resource-usage -save-temps=cwd \
    "$here/azgomoku.hip" -o "$out" -lpthread "$@" ) 2> "$log" || { cat "$log" >&2; rm -f "$log"; exit 1; }
asm="$(ls "$tmp"/*gfx950.s 2>/dev/null | head -1)"
if [ -z "$asm" ]; then echo "build.sh: no device assembly to check" >&2; rm -f "$out" "$log"; exit 1; fi
# (only the kernels that run beside matrix-dense waves: search, evaluation, caller-evaluated playouts;
#  the training kernels may use two-wide arithmetic)
bad="$(awk '/^_ZN3azg[^ ]*:/ { fn = $1 } /v_pk_(fma|mul|add)_f32/ && fn ~ /search_|eval_kernel|playout_/ { print fn " " $0 }' "$asm" | head -5)"
if [ -n "$bad" ]; then
    echo "$bad" >&2
    echo "build.sh: two-wide float32 instructions in a search / evaluation kernel (see above): refusing to ship this library" >&2
    rm -f "$out" "$log"; exit 1
fi
# compiler warnings, without the resource-usage remarks and their source excerpts
if grep -q "warning:" "$log"; then
    awk '/kernel-resource-usage/ { skip = 2; next } skip > 0 { skip--; next } { print }' "$log" >&2
fi
# The split-f16 search kernels for boards up to 8x8 run two workgroups per CU and are capped at 240
# VGPRs (azg_kernels.hpp, AZG_CAPPED_SEARCH_KERNEL), so that copy kernels can run beside them;
# this check reports a build in which the cap did not hold.
awk '/Function Name: _ZN3azg21search_kernel_f16_[678]x[678]ENS_8DevStateE/ { k = $0; sub(/.*Function Name: /, "", k); sub(/ .*/, "", k) }
     k != "" && / VGPRs: / { v = $0; sub(/.* VGPRs: /, "", v); sub(/ .*/, "", v);
         if (v + 0 > 240) printf("WARNING: %s uses %d VGPRs (> 240): copy kernels cannot run beside it\n", k, v);
         k = "" }' "$log" >&2
rm -f "$log"
echo "built $out"
