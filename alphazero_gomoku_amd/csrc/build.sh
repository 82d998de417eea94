#!/bin/sh
# Builds libazgomoku.so for gfx950 (MI355X).  -ffp-contract=off keeps the PUCT and
# running-mean arithmetic in the reference's operation order (no fused multiply-add).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../libazgomoku.so"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wall \
    "$here/azgomoku.hip" -o "$out" -lpthread "$@"
echo "built $out"
