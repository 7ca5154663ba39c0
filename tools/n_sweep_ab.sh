#!/bin/bash
# tools/n_sweep.sh across the large-N path's tile shapes: 128 x 128 (SX_TRMM_PT=8) against 128 x 64 (SX_TRMM_PT=4) and the
# launcher's own choice, one box.
cd "$(dirname "$0")/.."
for pt in 8 4 0; do
    echo "# SX_TRMM_PT=$pt"
    SX_TRMM_PT=$pt NS="${NS:-1000 1100 1200 1500 2000}" bash tools/n_sweep.sh
done
