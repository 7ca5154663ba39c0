"""Register / spill table of the kernels in one translation unit of libsxamd (no GPU needed):

    python tools/kernel_resources.py safe_exploration_amd/csrc/sx_rw_ns2.hip [substring of the kernel name]

Compiles the device side to assembly and prints, per kernel, VGPRs (arch + acc), AGPRs, spilled VGPRs, scratch bytes and
spilled SGPRs.  The register-resident rollout kernels (sx_rollout_rw.hpp) must show 0 scratch: sx_rw_launch.hpp's
rw_max_nrb is the largest n_pad / 16 per (n_s, n_u) for which they do.
"""
import os
import re
import subprocess
import sys
import tempfile


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-mllvm',
                               '-amdgpu-mfma-vgpr-form=1', '--cuda-device-only', '-S', src, '-o', out],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
    meta = text[text.index('amdhsa.kernels'):]
    print(f'{"kernel":70s} {"vgpr":>5s} {"agpr":>5s} {"vspill":>6s} {"scratch":>7s} {"sspill":>6s}')
    for block in re.split(r'\n  - \.agpr_count:', meta)[1:]:
        name = re.search(r'\.name:\s+(\S+)', block).group(1)
        if want not in name:
            continue
        demangled = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
        get = lambda key: re.search(r'\.' + key + r':\s+(\d+)', block).group(1)
        print(f'{demangled[-70:]:70s} {get("vgpr_count"):>5s} {block.split()[0]:>5s} {get("vgpr_spill_count"):>6s} '
              f'{get("private_segment_fixed_size"):>7s} {get("sgpr_spill_count"):>6s}')


if __name__ == '__main__':
    main()
