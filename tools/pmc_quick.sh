#!/bin/bash
# One quick PMC pass (counters given as arguments after the tag) over a bench command; prints per-kernel averages.
#   tools/pmc_quick.sh <tag> "<bench flags>" COUNTER...
set -euo pipefail
TAG=$1; FLAGS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline $FLAGS > $OUT.log 2>&1
find $OUT -name '*kernel_trace.csv' -delete
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
sums = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].split('(')[0][-40:]
        sums[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[k][row['Counter_Name']] += 1
for k in sums:
    if 'sx::' in k or 'kernel' in k:
        print(k, {c: round(sums[k][c] / cnt[k][c]) for c in sums[k]}, 'launches', max(cnt[k].values()))
PY
