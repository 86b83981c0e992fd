#!/usr/bin/env python3
"""What does the fused BatchNorm-backward epilogue of the f16x3 data gradient cost per launch, next to the reduce pass it replaces?

For the 1x1 data-gradient shapes of the step (dy [8, Cout, 128, 128] -> dx [8, Cin, 128, 128]): the plain launch, the launch with bnb (gate
recomputed from x / read from y), and `bn_backward_sums` (the separate reduction over dy and x).

  python tools/bnb_epilogue_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfst_amd import hip_ops as H      # noqa: E402


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    n, hw = 8, 128
    print(f'{"dy -> dx":>14s} {"plain":>8s} {"bnb x":>8s} {"bnb y":>8s} {"bnb bits":>8s} {"sums":>8s}   ms; fused costs (bnb - plain) against the separate pass (sums)')
    for co, ci in ((2048, 512), (1024, 256), (512, 2048), (256, 1024), (512, 128)):
        if not H.f16x3_eligible(co, ci, 1):
            continue
        dy = torch.randn(n, co, hw, hw, device='cuda')
        w = torch.randn(co, ci, 1, 1, device='cuda') * 0.02
        _, w4d, wa = H.pack_weight_f16x2(w, False, True)
        da = H.absmax(dy)
        pre = torch.randn(n, ci, hw, hw, device='cuda')
        g = torch.rand(ci, device='cuda') + 0.5
        b = torch.randn(ci, device='cuda') * 0.1
        mean, invstd, coef = H.bn_stats(pre, gamma=g, beta=b)
        y, mask = H.bn_apply(pre, mean, invstd, g, b, relu=True, residual=torch.zeros_like(pre), want_mask=True)
        out = torch.empty(n, ci, hw, hw, device='cuda')
        dg, db = torch.zeros(ci, device='cuda'), torch.zeros(ci, device='cuda')
        t0 = timeit(lambda: H.conv_dgrad_f16x3(dy, w4d, wa, da, ci, (hw, hw), 1, out=out))
        ok = ci % H.bnb_tile_rows(ci) == 0
        t1 = timeit(lambda: H.conv_dgrad_f16x3(dy, w4d, wa, da, ci, (hw, hw), 1, out=out, bnb=(pre, None, coef, True))) if ok else float('nan')
        t2 = timeit(lambda: H.conv_dgrad_f16x3(dy, w4d, wa, da, ci, (hw, hw), 1, out=out, bnb=(pre, y, coef, True))) if ok else float('nan')
        t2m = timeit(lambda: H.conv_dgrad_f16x3(dy, w4d, wa, da, ci, (hw, hw), 1, out=out, bnb=(pre, y, coef, True, mask))) if ok else float('nan')
        t3 = timeit(lambda: H.bn_backward_sums(out, pre, mean, invstd, g, b, dg, db))
        print(f'{co:6d} -> {ci:4d} {t0:8.3f} {t1:8.3f} {t2:8.3f} {t2m:8.3f} {t3:8.3f}   +{t1 - t0:.3f} / +{t2 - t0:.3f} / +{t2m - t0:.3f} vs {t3:.3f}', flush=True)


if __name__ == '__main__':
    main()
