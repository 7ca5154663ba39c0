#!/bin/bash
# Quick A/B builds of the register-resident rollout kernels: recompiles ONE translation unit (default sx_rw_ns2.hip) with
# extra flags and links it with the product build's other objects:
#     tools/rw_variant.sh <name> "<flags>" [unit]      ->  safe_exploration_amd/csrc/lib<name>.so
# Run the product build first (csrc/build.sh).  Compare on ONE box: tools/ab_libs.sh "libsxamd lib<name>" "2 5"
set -euo pipefail
cd "$(dirname "$0")/../safe_exploration_amd/csrc"
NAME=$1; FLAGS=${2:-}; UNIT=${3:-sx_rw_ns2}
mkdir -p build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -Wall -Wno-unused-function \
    $FLAGS -c $UNIT.hip -o build_var/${NAME}_$UNIT.o
OBJS=""
for o in build/*.o; do
    if [ "$(basename $o)" == "$UNIT.o" ]; then OBJS="$OBJS build_var/${NAME}_$UNIT.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o lib$NAME.so $OBJS
echo built lib$NAME.so
