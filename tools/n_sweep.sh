#!/bin/bash
# bench.py (config 2's problem: pendulum, 4096 particles, H = 15) over N_train across the path switches: all outputs at once
# (N <= 524) | output by output (<= 988) | three launches per step.  Prints ms per solve, the rollout time per iteration (sum
# of the path's kernels) and that time divided by N^2 (a flat last column = no cliff).  Non-zero status = no line.
for n in ${NS:-400 520 540 700 900 980 1000 1100 1200 1500 2000}; do
  python bench.py --config 2 --n-train $n --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
per_it = sum(v['avg_launch_us']*v['launches_timed'] for n,v in k.items() if n!='cem_rank_kernel')/ (max(v['launches_timed'] for n,v in k.items() if n!='cem_rank_kernel')/ (15 if 'trmm_reduce_kernel' in k else 1))
n=d['config']['n_train']
print(n, 'ms/solve', round(d['ms_per_step'],3), 'rollout us/iteration', round(per_it,1), 'us/N^2 x1e6', round(per_it/n/n*1e6,2), 'frac', round(d['roofline']['frac'],3), d['roofline']['kernel'], 'status', d['device_status'])"
done
