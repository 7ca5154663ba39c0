#!/bin/bash
# A/B of library builds on ONE box (kernel times move +-2 % from box to box): tools/ab_libs.sh "<lib names>" "<configs>" [reps]
# e.g. SX_OUT=libsx_a.so SX_EXTRA_FLAGS=-DSX_EPI=1 safe_exploration_amd/csrc/build.sh; gpurun -- 'bash tools/ab_libs.sh "libsxamd libsx_a" "2 5"'
LIBS=${1:-libsxamd}; CFGS=${2:-2}; REPS=${3:-2}
for rep in $(seq $REPS); do for v in $LIBS; do for c in $CFGS; do
  SX_LIB=$PWD/safe_exploration_amd/csrc/$v.so python bench.py --config $c --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/ab.json')); print(sys.argv[1], sys.argv[2], round(d['ms_per_step'],4), {k:round(v['avg_launch_us'],1) for k,v in d['kernels'].items()})" $v cfg$c
done; done; done
