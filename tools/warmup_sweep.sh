#!/bin/bash
# How long does the card take to reach its steady clock, and what do the timed region's length and the event stride do?
# The driver's command line (--steps 20 --warmup 5) against longer warm-ups, longer runs and sparser events, one box.
set -e
cd "$(dirname "$0")/.."
run() {
    python bench.py "$@" --no-cpu-baseline > gpurun_out/ws.json 2> gpurun_out/ws.err
    python - "$*" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ws.json').read().strip().splitlines()[-1])
r = d.get('roofline') or {'avg_launch_us': float('nan'), 'frac': float('nan'), 'launches_timed': 0}
print('%-50s' % sys.argv[1], 'ms_per_step %.4f' % d['ms_per_step'], 'kernel %.1f us (%d timed)' % (r['avg_launch_us'], r['launches_timed']), 'frac %.3f' % r['frac'])
PY
}
for rep in 1 2; do
    run --steps 20 --warmup 5
    run --steps 20 --warmup 50
    run --steps 20 --warmup 50 --profile-stride 16
    run --steps 20 --warmup 50 --no-kernel-timer
    run --steps 200 --warmup 50
    run --steps 200 --warmup 50 --profile-stride 3
    run --steps 2000 --warmup 50
    run --steps 2000 --warmup 50 --no-kernel-timer
done
