#!/bin/bash
# Fixed cost of one fused-rollout launch: bench.py at config 2 with H = 1, 2, 4, 8, 15 for the given kernel forms; the
# intercept of kernel time over H is what a launch spends outside its step loop.   tools/horizon_sweep.sh "rh stream"
FORMS=${1:-"rh stream"}
for f in $FORMS; do for h in 1 2 4 8 15; do
  SX_ROLLOUT=$f python bench.py --config 2 --horizon $h --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/hs.json 2> gpurun_out/hs.err || { tail -3 gpurun_out/hs.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/hs.json')); k=[v for n,v in d['kernels'].items() if 'rollout' in n][0]; print(sys.argv[1], 'H', sys.argv[2], round(k['avg_launch_us'],2), 'us per launch;', round(d['ms_per_step'],4), 'ms per solve')" $f $h
done; done
