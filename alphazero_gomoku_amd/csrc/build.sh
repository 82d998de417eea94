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
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wall -no-hip-rt \
    "$here/azgomoku.hip" -o "$out" -lpthread "$@"
echo "built $out"
