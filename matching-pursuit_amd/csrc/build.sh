#!/bin/bash
# Build libmpcore.so for MI355X (gfx950).  hipcc cross-compiles without a GPU.
# Output: matching-pursuit_amd/lib/libmpcore.so  (git-ignored; ships to the GPU box with gpurun)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(dirname "$(dirname "$HERE")")"
OUT="$(dirname "$HERE")/lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -ffp-contract=off: the residual update is two roundings (r - d*g), never an fma
"$HIPCC" --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 \
    -I"$ROOT/include" -o "$OUT/libmpcore.so" "$HERE/mpcore.hip" "$@"
echo "built $OUT/libmpcore.so"
