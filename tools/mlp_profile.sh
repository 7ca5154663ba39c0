#!/bin/bash
# rocprofv3 over tools/dropout_probe.py on the GPU box: --stats, then one PMC pass; keeps only the rows of the MC-dropout kernels.
#   tools/mlp_profile.sh <tag> [P]      -> gpurun_out/<tag>_mlp_stats.csv, gpurun_out/<tag>_mlp_pmc.csv
set -euo pipefail
TAG=$1; export P=${2:-16384}; export ITERS=${ITERS:-5}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mlp_stats /tmp/mlp_pmc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mlp_stats -- python3 $ROOT/tools/dropout_probe.py > $OUT/${TAG}_mlp_stats.log 2>&1
f=$(find /tmp/mlp_stats -name '*kernel_stats.csv' | head -1)
(head -1 $f; grep mlp $f) > $OUT/${TAG}_mlp_stats.csv
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv \
    -d /tmp/mlp_pmc -- python3 $ROOT/tools/dropout_probe.py > $OUT/${TAG}_mlp_pmc.log 2>&1
f=$(find /tmp/mlp_pmc -name '*counter_collection.csv' | head -1)
(head -1 $f; grep mlp $f) > $OUT/${TAG}_mlp_pmc.csv
echo "profiled"
