#!/usr/bin/env python3
"""Can an HBM-bound kernel of the train step hide beside an f16x3 matrix kernel on this chip?  Two HIP streams, one kernel family each.

For each (GEMM, streaming) pair: time R back-to-back launches of each alone, then both queued on their own streams at once, and report
t_both / (t_gemm + t_stream)  (1.0 = serialised, max(t_g, t_s) / (t_g + t_s) = perfect overlap).  The f16x3 GEMM holds 200 VGPRs x 8 waves and
96 KB of LDS per CU: 96 registers per lane and 64 KB stay free on every SIMD / CU, which a streaming kernel's waves may take.

  python tools/overlap_microbench.py [--gemm fprop2048,fprop256,wgrad] [--stream-kernels bn_apply,bn_bwd] [--prio none,gemm,stream] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfst_amd import hip_ops as H      # noqa: E402


def make_gemm(kind, n=8, hw=128):
    cin, cout = {'fprop2048': (2048, 512), 'fprop256': (256, 1024), 'fprop512': (512, 2048), 'wgrad': (2048, 512)}[kind]
    x = torch.randn(n, cin, hw, hw, device='cuda').relu_()
    xa = H.absmax(x)
    flops = 2.0 * n * cin * cout * hw * hw
    if kind == 'wgrad':
        dy = torch.randn(n, cout, hw, hw, device='cuda')
        da = H.absmax(dy)
        dw = torch.zeros(cout, cin, 1, 1, device='cuda')
        return lambda: H.conv_wgrad_f16x3_(dw, x, dy, xa, da), flops
    w = torch.randn(cout, cin, 1, 1, device='cuda') * 0.02
    w4f, _, wa = H.pack_weight_f16x2(w, True, False)
    out = torch.empty(n, cout, hw, hw, device='cuda')
    return lambda: H.conv_fprop_f16x3(x, w4f, wa, xa, cout, 1, out=out, want_stats=True), flops


def make_stream_kernel(kind, n=8, c=1024, hw=128):
    x = torch.randn(n, c, hw, hw, device='cuda')
    g = torch.rand(c, device='cuda') + 0.5
    b = torch.randn(c, device='cuda') * 0.1
    mean, invstd = H.bn_stats(x)
    if kind == 'bn_apply':
        y = torch.empty_like(x)
        return lambda: H.bn_apply(x, mean, invstd, g, b, relu=True, out=y), 4.0 * x.numel() * 2
    if kind == 'bn_apply_res':
        y = torch.empty_like(x)
        r = torch.randn_like(x)
        return lambda: H.bn_apply(x, mean, invstd, g, b, relu=True, residual=r, out=y), 4.0 * x.numel() * 3
    if kind == 'bn_bwd':
        dy = torch.randn_like(x)
        dx = torch.empty_like(x)
        dg, db = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
        return lambda: H.bn_backward(dy, None, x, mean, invstd, g, dg, db, relu=True, dx=dx, beta=b), 4.0 * x.numel() * 5
    raise KeyError(kind)


def timed(fn_pairs, reps):
    """fn_pairs: [(stream, fn)]: every fn is queued `reps` times on its stream; wall time from a common start event to the last end"""
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    ends = []
    for s, fn in fn_pairs:
        s.wait_event(start)
    for _ in range(reps):                  # interleave the host-side launches so that neither queue runs dry
        for s, fn in fn_pairs:
            with torch.cuda.stream(s):
                fn()
    for s, fn in fn_pairs:
        e = torch.cuda.Event(enable_timing=True)
        e.record(s)
        ends.append(e)
    torch.cuda.synchronize()
    return max(start.elapsed_time(e) for e in ends) / reps, [start.elapsed_time(e) / reps for e in ends]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gemm', default='fprop2048,fprop512,fprop256,wgrad')
    ap.add_argument('--stream-kernels', default='bn_apply,bn_apply_res,bn_bwd')
    ap.add_argument('--prio', default='none,gemm,stream')
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    for gk in args.gemm.split(','):
        gemm, flops = make_gemm(gk)
        for sk in args.stream_kernels.split(','):
            sfn, nbytes = make_stream_kernel(sk)
            for prio in args.prio.split(','):
                sg = torch.cuda.Stream(priority=-1 if prio == 'gemm' else 0)
                ss = torch.cuda.Stream(priority=-1 if prio == 'stream' else 0)
                for s in (sg, ss):         # the workspaces (statistics scratch, amax arena) are per stream: warm both
                    with torch.cuda.stream(s):
                        for _ in range(2):
                            gemm(); sfn()
                tg, _ = timed([(sg, gemm)], args.reps)
                ts, _ = timed([(ss, sfn)], args.reps)
                tb, each = timed([(sg, gemm), (ss, sfn)], args.reps)
                ideal = max(tg, ts) / (tg + ts)
                print(f'{gk:10s} + {sk:12s} prio={prio:6s}  gemm {tg:6.3f} ms ({flops / tg / 1e9:6.1f} TF-eq)  stream {ts:6.3f} ms '
                      f'({nbytes / ts / 1e6:5.0f} GB/s)  both {tb:6.3f} ms [{each[0]:.3f} | {each[1]:.3f}]  ratio {tb / (tg + ts):.3f} '
                      f'(ideal {ideal:.3f})', flush=True)


if __name__ == '__main__':
    main()
