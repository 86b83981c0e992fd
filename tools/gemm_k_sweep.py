"""Per-tile overhead of the f16x3 implicit GEMM: the same 1x1 convolution (8 x 128^2 pixels) over contraction lengths K = 64 .. 2048.
1 / rate is linear in 1 / K: the intercept is the time a tile spends outside its K loop (prologue, epilogue), in K=32 steps.
  python tools/gemm_k_sweep.py [--m 512] [--wino]        (PFST_HIP_LIB=<other build> for a same-box A/B)"""
import argparse
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfst_amd import hip_ops as H      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--m', default='512,2048')
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--stats', type=int, default=0, help='1: with the fused BatchNorm statistics (as every conv -> BN layer runs), 2: + min / max partials')
    ap.add_argument('--bnl', action='store_true', help='the normalising form: the input is a pre-BN tensor, normalised between load and split (K <= 2048)')
    args = ap.parse_args()
    n, hw = 8, 128
    for m in [int(v) for v in args.m.split(',')]:
        pts = []
        for k in (64, 128, 256, 512, 1024, 2048):
            x = torch.randn(n, k, hw, hw, device='cuda')
            w = torch.randn(m, k, 1, 1, device='cuda') * 0.05
            w4f, _, wa = H.pack_weight_f16x2(w, True, False)
            xa = H.absmax(x)
            out = torch.empty(n, m, hw, hw, device='cuda')
            coef = None
            if args.bnl and H.conv_fprop_bnl_ok(k, m, 1):
                coef = torch.stack([torch.zeros(k), torch.ones(k), torch.rand(k) + 0.5, torch.randn(k) * 0.1], 1).cuda().contiguous()
            for _ in range(3):
                H.conv_fprop_f16x3(x, w4f, wa, xa, m, 1, out=out, want_stats=args.stats > 0, want_minmax=args.stats > 1, bnl=coef)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.reps):
                H.conv_fprop_f16x3(x, w4f, wa, xa, m, 1, out=out, want_stats=args.stats > 0, want_minmax=args.stats > 1, bnl=coef)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / args.reps
            tf = 2.0 * n * m * k * hw * hw / ms / 1e9
            pts.append((k // 32, ms))
            print(f'M={m:5d} K={k:5d}  {ms:7.3f} ms  {tf:6.1f} TF/s-eq', flush=True)
        # least squares ms = a + b * steps over K >= 128
        xs = [p[0] for p in pts[1:]]
        ys = [p[1] for p in pts[1:]]
        mx, my = sum(xs) / len(xs), sum(ys) / len(ys)
        b = sum((u - mx) * (v - my) for u, v in zip(xs, ys)) / sum((u - mx) ** 2 for u in xs)
        a = my - b * mx
        print(f'M={m:5d}: per-tile overhead = {a / b:.1f} K-steps; asymptotic rate {2.0 * n * m * 32 * hw * hw / b / 1e9:.0f} TF/s-eq', flush=True)


if __name__ == '__main__':
    main()
