#!/bin/bash
# Builds libsxamd.so (gfx950 code object, host code for this machine) next to the sources.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${SX_OUT:-libsxamd.so}
SRC="sx_kernels.hip"
# rebuild only when a source is newer than the library
if [ -f "$OUT" ] && [ -z "$(find . ../../include -newer "$OUT" \( -name '*.hip' -o -name '*.hpp' -o -name '*.h' -o -name 'build.sh' \) | head -1)" ]; then
    exit 0
fi
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -mllvm -amdgpu-mfma-vgpr-form=1 -Wall -Wno-unused-function ${SX_EXTRA_FLAGS:-} -o "$OUT.tmp" $SRC
mv "$OUT.tmp" "$OUT"
