#!/bin/sh
# Experiment builds: sh build_variant.sh NAME [-D...]  ->  abx/libazg_NAME.so with the 8x8 split-f16
# kernels only (AZG_ONLY=15 in the environment: the 15x15 kernels only); select it with
# AZG_LIBRARY=abx/libazg_NAME.so.  Prints the register use of the search kernel.
# AZG_SLP=1 in the environment leaves the SLP vectoriser on (the shipped build turns it off, see build.sh);
# AZG_KEEP=1 keeps the device assembly beside the library (abx/NAME-hip-amdgcn-*.s).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
name="$1"; shift
mkdir -p "$here/../../abx"
out="$here/../../abx/libazg_$name.so"
log="$here/../../abx/build_$name.log"
slp="-fno-slp-vectorize"; [ "${AZG_SLP:-0}" = 1 ] && slp=""
keep=""; [ "${AZG_KEEP:-0}" = 1 ] && keep="-save-temps=obj"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off $slp $keep -fPIC -shared -std=c++17 -no-hip-rt -DAZG_ONLY${AZG_ONLY:-8} \
    -Rpass-analysis=kernel-resource-usage "$here/azgomoku.hip" -o "$out" -lpthread "$@" 2> "$log" || { tail -30 "$log" >&2; exit 1; }
[ "${AZG_QUIET:-0}" = 1 ] || grep -A12 -E "Function Name: _ZN3azg(21search_kernel_f16_8x8|18search_dual_kernel|13search_kernelILi8ELi[0-9]ELi0ELb1)" "$log" | grep -E "VGPRs:|Spill|LDS Size|Occupancy" | sed "s/^.*remark: [^ ]* //" | tr '\n' ' '
echo " -> $out"
