#!/bin/bash
# What does the sampled kernel timer cost, and what does it read?  Events attached to the launch (hipExtLaunchKernelGGL,
# the default) against two hipEventRecord around it (SX_PROF_RECORD=1), at several strides, one box.
set -e
cd "$(dirname "$0")/.."
run() {
    python bench.py "$@" --no-cpu-baseline > gpurun_out/ws.json 2> gpurun_out/ws.err
    python - "SX_PROF_RECORD=${SX_PROF_RECORD:-0} $*" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ws.json').read().strip().splitlines()[-1])
r = d.get('roofline') or {'avg_launch_us': float('nan'), 'frac': float('nan'), 'launches_timed': 0}
k = d.get('kernels') or {}
rank = k.get('cem_rank_kernel', {}).get('avg_launch_us', float('nan'))
print('%-62s' % sys.argv[1], 'ms_per_step %.4f' % d['ms_per_step'], 'rollout %.1f us (%d timed)' % (r['avg_launch_us'], r['launches_timed']), 'rank %.1f us' % rank, 'frac %.3f' % r['frac'])
PY
}
for rep in 1 2; do
    run --steps 200 --warmup 50 --no-kernel-timer
    for m in 0 1; do
        export SX_PROF_RECORD=$m
        run --steps 200 --warmup 50 --profile-stride 1
        run --steps 200 --warmup 50 --profile-stride 3
        run --steps 200 --warmup 50 --profile-stride 16
        run --steps 20 --warmup 5
    done
    unset SX_PROF_RECORD
done
