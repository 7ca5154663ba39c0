#!/bin/bash
# One BASELINE config through rocprofv3 on the GPU box: --stats (kernel trace) and the three PMC passes, each its own run.
#   tools/profile_config.sh <cfg> <tag> [bench flags]      -> gpurun_out/<tag>_cfg<cfg>_{stats,fetch,write,sq}/ + logs
# Summaries to commit: the *_kernel_stats.csv of the stats run and tools/pmc_summary.py over the three PMC directories.
set -euo pipefail
CFG=$1; TAG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --config $CFG --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cfg${CFG}_stats -- $B > $OUT/${TAG}_cfg${CFG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_cfg${CFG}_fetch -- $B > $OUT/${TAG}_cfg${CFG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_cfg${CFG}_write -- $B > $OUT/${TAG}_cfg${CFG}_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY \
    --kernel-trace --output-format csv -d $OUT/${TAG}_cfg${CFG}_sq -- $B > $OUT/${TAG}_cfg${CFG}_sq.log 2>&1
# the raw traces are large: keep the summaries only
find $OUT/${TAG}_cfg${CFG}_stats -name '*kernel_trace.csv' -delete
for d in fetch write sq; do find $OUT/${TAG}_cfg${CFG}_$d -name '*kernel_trace.csv' -delete; done
echo "profiled cfg $CFG"
