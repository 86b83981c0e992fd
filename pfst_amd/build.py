"""Build libpfst_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

`python -m pfst_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libpfst_hip.so')
SOURCES = ['conv_mfma.hip', 'conv_igemm_q.hip', 'conv_wgrad_q.hip', 'conv_winograd.hip', 'conv_split.hip', 'conv_f16x3.hip', 'dwconv.hip', 'bn.hip', 'spatial.hip', 'loss.hip', 'pfgst_loss.hip', 'optim.hip', 'strong_aug.hip', 'api.cpp']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-Wno-unused-result']
# No packed-fp32 (v_pk_*_f32) code in the streaming kernels: round 5 found the depthwise backward kernel (dwconv3x3_kernel<3, true>: a packed add
# on a pair of weight-gradient accumulators) returning WRONG sums -- one accumulator of a few channels off by O(1) -- whenever it ran on the
# same CUs as the f16x3 weight-gradient kernel of the side stream, and only then (alone, or beside a rocBLAS GEMM: exact; tools/race_probe.py,
# profiles/r05_packed_fp32_corruption.txt).  Without the SLP vectoriser clang forms no <2 x float> operations, the kernels are exact under any
# co-residency and 5-14 % FASTER (they are LDS / HBM bound, the packed forms bought nothing).  The matrix kernels keep the default: their
# packed epilogue arithmetic is worth 9 ms per step and is pinned bit for bit by the deterministic-mode test under stream overlap.
NO_SLP = {'dwconv.hip', 'bn.hip', 'spatial.hip', 'loss.hip', 'pfgst_loss.hip', 'optim.hip', 'strong_aug.hip', 'conv_winograd.hip'}


def _flags(src_name):
    return FLAGS + (['-fno-slp-vectorize'] if src_name in NO_SLP else [])


def _deps(src, seen=None):
    """src + every header it reaches through `#include "..."` (csrc/ and include/), so that editing a shared header
    (conv_epilogue.h, common.h, pfst_hip.h) rebuilds all of its includers"""
    import re
    seen = set() if seen is None else seen
    if src in seen or not os.path.exists(src):
        return seen
    seen.add(src)
    for inc in re.findall(r'^\s*#\s*include\s*"([^"]+)"', open(src).read(), flags=re.M):
        for base in (os.path.dirname(src), CSRC, os.path.join(HERE, '..', 'include')):
            cand = os.path.normpath(os.path.join(base, inc))
            if os.path.exists(cand):
                _deps(cand, seen)
                break
    return seen


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in _deps(src))


CPU_LIB = os.path.join(HERE, 'libpfst_cpu.so')


def build_cpu(force=False, verbose=True):
    """libpfst_cpu.so: the data pipeline's pixel kernels (csrc/pipeline_cpu.c), plain C with gcc.  -ffp-contract=off: no fused
    multiply-adds, so every float operation rounds like the NumPy expression it mirrors (bit-identical results)."""
    src = os.path.join(CSRC, 'pipeline_cpu.c')
    if force or not os.path.exists(CPU_LIB) or os.path.getmtime(src) > os.path.getmtime(CPU_LIB):
        cmd = [os.environ.get('CC', 'gcc'), '-O3', '-msse4.1', '-fPIC', '-shared', '-std=c99', '-ffp-contract=off', '-fno-fast-math', '-o', CPU_LIB, src, '-lm']
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return CPU_LIB


def build(force=False, verbose=True):
    build_cpu(force, verbose)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.rsplit('.', 1)[0] + '.o')
        objs.append(obj)
        if force or _stale(obj, src):
            cmd = [hipcc] + _flags(s) + (['-x', 'hip'] if s.endswith('.cpp') else []) + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f'--- {s} failed\n{out.decode()}\n')
        elif verbose and out.strip():
            print(out.decode())
    if failed:
        raise RuntimeError('hipcc failed')
    if force or procs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, extra_flags):
    """a second build of the library with extra compiler flags (diagnostic -D switches, alternative tunings) as
    ab_libs/libpfst_hip_<name>.so -- for same-box A/B runs through PFST_HIP_LIB (tools/ab_lib.sh); git-ignored, travels with gpurun"""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    root = os.path.join(HERE, '..', 'ab_libs')
    objdir = os.path.join(root, name)
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.rsplit('.', 1)[0] + '.o')
        objs.append(obj)
        cmd = [hipcc] + _flags(s) + list(extra_flags) + (['-x', 'hip'] if s.endswith('.cpp') else []) + ['-c', src, '-o', obj]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'{s}: {out.decode()}')
    lib = os.path.normpath(os.path.join(root, f'libpfst_hip_{name}.so'))
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    return lib


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--variant':
        print(build_variant(sys.argv[2], sys.argv[3:]))
    else:
        build(force='--force' in sys.argv)
        print(LIB)
