#!/bin/bash
# A/B of the three forms of the fused rollout kernel on ONE box: tools/ab_forms.sh "<forms>" "<configs>" [reps]
# forms: rh (8 waves, W partly resident) | rw (4 waves, W in registers) | stream (W from L2)
FORMS=${1:-"rh rw stream"}; CFGS=${2:-2}; REPS=${3:-2}
for rep in $(seq $REPS); do for f in $FORMS; do for c in $CFGS; do
  SX_ROLLOUT=$f python bench.py --config $c --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(sys.argv[1], sys.argv[2], round(d['ms_per_step'],4), {k:round(v['avg_launch_us'],1) for k,v in d['kernels'].items()})" $f cfg$c
done; done; done
