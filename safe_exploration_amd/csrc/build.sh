#!/bin/bash
# Builds libsxamd.so (gfx950 code object, host code for this machine) next to the sources.  The translation units are
# compiled in parallel (objects under build/, which is git-ignored) and linked into one shared library.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${SX_OUT:-libsxamd.so}
SRCS="sx_kernels.hip sx_rw_ns1.hip sx_rw_ns2.hip sx_rw_ns3.hip sx_rw_ns4.hip"
# rebuild only when a source is newer than the library
if [ -f "$OUT" ] && [ -z "$(find . ../../include -newer "$OUT" \( -name '*.hip' -o -name '*.hpp' -o -name '*.h' -o -name 'build.sh' \) | head -1)" ]; then
    exit 0
fi
OBJ=${SX_OBJDIR:-build}
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -Wall -Wno-unused-function ${SX_EXTRA_FLAGS:-}"
pids=()
objs=()
for src in $SRCS; do
    obj="$OBJ/${src%.hip}.o"
    objs+=("$obj")
    "$HIPCC" $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
done
for pid in "${pids[@]}"; do wait "$pid"; done
"$HIPCC" --offload-arch=gfx950 -fPIC -shared -o "$OUT.tmp" "${objs[@]}"
mv "$OUT.tmp" "$OUT"
