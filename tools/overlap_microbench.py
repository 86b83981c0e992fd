"""Can an HBM-bound kernel of the train step hide beside an MFMA-bound one on this chip?  Two HIP streams, one kernel family each.

For each (GEMM, streaming) pair: time R back-to-back launches of each alone, then both queued on their own streams at once, and
report  t_both / (t_gemm + t_stream)  (1.0 = serialised, max(t_g, t_s) / (t_g + t_s) = perfect overlap).
Knobs: stream priority of either side, occupancy caps of the GEMM kernels (unused dynamic LDS: PFST_WGRAD_LDS_PAD via
ops.set_wgrad_lds_pad, PFST_SPLIT_LDS_PAD / PFST_IGEMM_LDS_PAD at process start).

  python tools/overlap_microbench.py [--gemm f32|split|wgrad|wgrad_split] [--prio gemm|stream|none] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfst_amd import hip_ops as H      # noqa: E402


def make_gemm(kind, n=8, cin=2048, cout=512, hw=128):
    x = torch.randn(n, cin, hw, hw, device='cuda')
    w = torch.randn(cout, cin, 1, 1, device='cuda') * 0.02
    if kind == 'f32':
        wf, _ = H.pack_weight(w, True, False)
        out = torch.empty(n, cout, hw, hw, device='cuda')
        return lambda: H.conv_fprop(x, wf, cout, 1, out=out), 2.0 * n * cin * cout * hw * hw
    if kind == 'split':
        w6, _ = H.pack_weight_split(w, True, False)
        out = torch.empty(n, cout, hw, hw, device='cuda')
        return lambda: H.conv_fprop_split(x, w6, cout, 1, out=out), 2.0 * n * cin * cout * hw * hw
    dy = torch.randn(n, cout, hw, hw, device='cuda')
    dw = torch.zeros(cout, cin, 1, 1, device='cuda')
    if kind == 'wgrad':
        return lambda: H.conv_wgrad_(dw, x, dy, 1), 2.0 * n * cin * cout * hw * hw
    if kind == 'wgrad_split':
        return lambda: H.conv_wgrad_split_(dw, x, dy, 1), 2.0 * n * cin * cout * hw * hw
    raise KeyError(kind)


def make_stream_kernel(kind, n=8, c=1024, hw=128):
    x = torch.randn(n, c, hw, hw, device='cuda')
    g = torch.rand(c, device='cuda') + 0.5
    b = torch.randn(c, device='cuda') * 0.1
    mean, invstd = H.bn_stats(x)
    if kind == 'bn_apply':
        y = torch.empty_like(x)
        return lambda: H.bn_apply(x, mean, invstd, g, b, relu=True, out=y), 4.0 * x.numel() * 2
    if kind == 'bn_bwd':
        dy = torch.randn_like(x)
        dx = torch.empty_like(x)
        dg, db = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
        return lambda: H.bn_backward(dy, None, x, mean, invstd, g, dg, db, relu=True, dx=dx, beta=b), 4.0 * x.numel() * 5
    if kind == 'dwconv':
        w = torch.randn(c, 1, 3, 3, device='cuda')
        y = torch.empty_like(x)
        return lambda: H.dwconv(x, w, 12, out=y), 4.0 * x.numel() * 2
    raise KeyError(kind)


def timed(fn_pairs, reps):
    """fn_pairs: [(stream, fn)]: every fn is queued `reps` times on its stream; wall time from a common start event to the last end"""
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    ends = []
    for s, fn in fn_pairs:
        s.wait_event(start)
    # interleave the host-side launches so that neither queue runs dry
    for _ in range(reps):
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
    ap.add_argument('--gemm', default='f32,split,wgrad,wgrad_split')
    ap.add_argument('--stream-kernels', default='bn_bwd,bn_apply,dwconv')
    ap.add_argument('--prio', default='none,gemm,stream')
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--wgrad-pad', type=int, default=None)
    args = ap.parse_args()
    if args.wgrad_pad is not None:
        H.set_wgrad_lds_pad(args.wgrad_pad)
    print('env caps:', {k: v for k, v in os.environ.items() if 'LDS_PAD' in k}, 'wgrad pad', args.wgrad_pad, flush=True)
    for gk in args.gemm.split(','):
        gemm, flops = make_gemm(gk)
        for sk in args.stream_kernels.split(','):
            sfn, nbytes = make_stream_kernel(sk)
            for prio in args.prio.split(','):
                sg = torch.cuda.Stream(priority=-1 if prio == 'gemm' else 0)
                ss = torch.cuda.Stream(priority=-1 if prio == 'stream' else 0)
                for _ in range(3):
                    gemm(); sfn()
                tg, _ = timed([(sg, gemm)], args.reps)
                ts, _ = timed([(ss, sfn)], args.reps)
                tb, each = timed([(sg, gemm), (ss, sfn)], args.reps)
                ideal = max(tg, ts) / (tg + ts)
                print(f'{gk:12s} + {sk:9s} prio={prio:6s}  gemm {tg:6.3f} ms ({flops / tg / 1e9:6.1f} TF/s)  stream {ts:6.3f} ms '
                      f'({nbytes / ts / 1e6:5.0f} GB/s)  both {tb:6.3f} ms [{each[0]:.3f} | {each[1]:.3f}]  ratio {tb / (tg + ts):.3f} '
                      f'(ideal {ideal:.3f})', flush=True)


if __name__ == '__main__':
    main()
