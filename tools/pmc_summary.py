"""Fold the rocprofv3 --pmc passes of bench.py into profiles/r<NN>_pmc_cfg<N>.json (per-launch averages per kernel).

On the GPU box (counters in passes of their own, never together with a runtime trace; tools/profile_config.sh does all
of it for one config):

    cd /tmp && export TMPDIR=/tmp
    B="python3 $GRAFT_REPO_ROOT/bench.py --config N --steps 2 --warmup 1 --no-cpu-baseline"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- $B
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- $B
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU \\
              SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -- $B

then:  python tools/pmc_summary.py --workload "cfgN N_train=.. H=.. P=.. E=.." DIR... > profiles/rNN_pmc_cfgN.json
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = ['cem_rollout_rh_kernel', 'cem_rollout_rw_kernel', 'cem_rollout_kernel', 'cem_rank_kernel', 'cem_rank_count_kernel', 'trmm_reduce_kernel', 'kstar_big_kernel', 'step_big_kernel']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', required=True, help='the key bench.py matches: "cfgN N_train=.. H=.. P=.. E=.."')
    ap.add_argument('--command', default='')
    ap.add_argument('dirs', nargs='+')
    args = ap.parse_args()
    # a kernel may run in several grid sizes (config 4's warm start predicts ONE point through the large-N kernels, 20 tiny
    # launches per solve beside the 160 of the rollout): average the launches of the LARGEST grid only -- the ones bench.py
    # times and prices -- and say how many others were left out
    by_grid = defaultdict(lambda: defaultdict(lambda: defaultdict(list)))
    full_names = {}
    for d in args.dirs:
        for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row['Kernel_Name']
                    for key in KERNELS:
                        if key in name:
                            by_grid[key][int(row['Grid_Size'])][row['Counter_Name']].append(float(row['Counter_Value']))
                            full_names[key] = name
    sums = defaultdict(lambda: defaultdict(float))
    counts = defaultdict(lambda: defaultdict(int))
    other_grids = {}
    for key, grids in by_grid.items():
        top = max(grids)
        for cname, vals in grids[top].items():
            sums[key][cname] = sum(vals)
            counts[key][cname] = len(vals)
        left = {str(g): max(len(v) for v in c.values()) for g, c in grids.items() if g != top}
        if left:
            other_grids[key] = {'grid_size_averaged': top, 'launches_left_out_by_grid_size': left}
    per_launch = {k: {c: sums[k][c] / counts[k][c] for c in sorted(sums[k])} for k in sums}
    launches = {k: max(counts[k].values()) for k in counts}
    out = {
        'command': args.command or 'rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py --config N '
                                   '--steps 2 --warmup 1 --no-cpu-baseline (three separate passes: FETCH_SIZE | WRITE_SIZE | '
                                   'SQ_*); tools/pmc_summary.py',
        'workload': args.workload,
        'kernel_names': full_names,
        'launches_counted': launches,
        'other_grid_sizes': other_grids,
        'per_launch_averages': per_launch,
        'notes': [
            'FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for wide '
            'coalesced reads on gfx950; Infinity-Cache hits are counted, not excluded',
            'SQ_VALU_MFMA_BUSY_CYCLES = 64 x SQ_INSTS_MFMA exactly: one v_mfma_f64_16x16x4_f64 occupies the pipe for 64 cycles',
            'mfma_pipe_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)',
        ],
        'hbm_traffic_bytes_per_launch': {}, 'mfma_pipe_busy_fraction': {},
    }
    for k, r in per_launch.items():
        if 'FETCH_SIZE' in r and 'WRITE_SIZE' in r:
            out['hbm_traffic_bytes_per_launch'][k] = (2 * r['FETCH_SIZE'] + r['WRITE_SIZE']) * 1024
        if r.get('SQ_VALU_MFMA_BUSY_CYCLES') and r.get('GRBM_GUI_ACTIVE'):
            out['mfma_pipe_busy_fraction'][k] = r['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * r['GRBM_GUI_ACTIVE'] / 8)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()
