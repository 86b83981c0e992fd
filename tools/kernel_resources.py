#!/usr/bin/env python3
"""Register / LDS / spill figures of every kernel of one csrc file, from the code object's metadata (cross-compiles, no GPU):
    python tools/kernel_resources.py conv_f16x3.hip [filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = os.path.join(ROOT, 'pfst_amd', 'csrc', sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    extra = sys.argv[3:]
    with tempfile.TemporaryDirectory() as d:
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-Wno-unused-result',
                        '-c', src, '-o', os.path.join(d, 'x.o'), '--save-temps=obj'] + extra, cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(d) if f.endswith('gfx950.s')][0]
        text = open(os.path.join(d, asm)).read()
    meta = text[text.index('amdhsa.kernels:'):]
    for blk in meta.split('  - .agpr_count:')[1:]:
        blk = '.agpr_count:' + blk
        f = {k: v for k, v in re.findall(r'\.(\w+):\s+(\S+)', blk)}
        name = subprocess.run(['c++filt', f.get('name', '?')], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0].replace('void ', '')
        if flt in name:
            print(f"{name:60s} vgpr {f.get('vgpr_count'):>4s} agpr {f.get('agpr_count'):>4s} sgpr {f.get('sgpr_count'):>4s} "
                  f"spill v {f.get('vgpr_spill_count'):>3s} s {f.get('sgpr_spill_count'):>3s}  lds {f.get('group_segment_fixed_size'):>6s}  scratch {f.get('private_segment_fixed_size')}")


if __name__ == '__main__':
    main()
