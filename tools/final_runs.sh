#!/bin/bash
# The round's reference numbers on ONE box: the whole GPU suite, smoke(), bench.py at every BASELINE config (plus the driver's
# command line and the ARD variant of config 2) and the three forms of the fused rollout kernel side by side.
#   gpurun --timeout 1200 -- 'bash tools/final_runs.sh r03'        -> gpurun_out/<tag>_*.json|txt (copy into profiles/)
set -e
TAG=${1:-r03}
cd "$(dirname "$0")/.."
timeout -k 10 200 python tools/rw_repro.py > gpurun_out/${TAG}_repro.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/${TAG}_pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/${TAG}_pytest_gpu.txt; exit 1; }
tail -2 gpurun_out/${TAG}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1 || { tail gpurun_out/${TAG}_smoke.txt; exit 1; }
tail -1 gpurun_out/${TAG}_smoke.txt
for c in 2 1 3 5 4; do
    python bench.py --config $c > gpurun_out/${TAG}_bench_cfg$c.json 2> gpurun_out/${TAG}_bench_cfg$c.err || { tail -5 gpurun_out/${TAG}_bench_cfg$c.err; exit 1; }
    echo "cfg $c done"
done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_cfg2_driver.json 2> gpurun_out/${TAG}_bench.err
python bench.py --ard > gpurun_out/${TAG}_bench_cfg2_ard.json 2> gpurun_out/${TAG}_bench.err
bash tools/ab_forms.sh "rh rw stream" "2" 2 > gpurun_out/${TAG}_ab_forms.txt 2>&1
cat gpurun_out/${TAG}_ab_forms.txt
