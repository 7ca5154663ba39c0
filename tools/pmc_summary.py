"""Fold the rocprofv3 --pmc passes of bench.py into profiles/<name>.json (per-launch averages of the two kernels).

On the GPU box (counters in passes of their own, never together with a runtime trace):

    cd /tmp && export TMPDIR=/tmp
    B="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- $B
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- $B
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU \\
              SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -- $B

then here:  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq > profiles/rNN_pmc_summary.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

# (needle: the all-outputs-at-once instantiation of the rollout kernel; its third template argument is BYOUT = false)
KERNELS = {'cem_rollout_kernel<2,1>': 'cem_rollout_kernel<2, 1, false>', 'cem_rank_kernel<4>': 'cem_rank_kernel<4>'}


def main(dirs):
    sums = defaultdict(lambda: defaultdict(float))
    counts = defaultdict(lambda: defaultdict(int))
    for d in dirs:
        for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row['Kernel_Name']
                    for key, needle in KERNELS.items():
                        if needle.replace(' ', '') in name.replace(' ', ''):
                            sums[key][row['Counter_Name']] += float(row['Counter_Value'])
                            counts[key][row['Counter_Name']] += 1
    per_launch = {k: {c: sums[k][c] / counts[k][c] for c in sorted(sums[k])} for k in sums}
    out = {
        'command': 'rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline '
                   '(three separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*); tools/pmc_summary.py',
        'workload': 'cfg2 pendulum N_train=200 H=15 P=4096',
        'per_launch_averages': per_launch,
        'notes': [
            'FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for wide '
            'coalesced reads on gfx950',
            'SQ_VALU_MFMA_BUSY_CYCLES = 64 x SQ_INSTS_MFMA exactly: one v_mfma_f64_16x16x4_f64 occupies the pipe for 64 cycles',
            'mfma_pipe_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)',
        ],
    }
    r = per_launch.get('cem_rollout_kernel<2,1>', {})
    if 'FETCH_SIZE' in r and 'WRITE_SIZE' in r:
        out['hbm_traffic_bytes_per_launch'] = {'cem_rollout_kernel<2,1>': (2 * r['FETCH_SIZE'] + r['WRITE_SIZE']) * 1024}
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in r and 'GRBM_GUI_ACTIVE' in r:
        out['mfma_pipe_busy_fraction'] = r['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * r['GRBM_GUI_ACTIVE'] / 8)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main(sys.argv[1:])
