#!/usr/bin/env python3
"""The second pass of BatchNorm backward alone (the sums given as epilogue partials, so no reduction pass runs) and both passes, at the
workload's tensor sizes, cold cache (384 MB written between repetitions).  For A/B builds (-DPFST_BN_BWD_APPLY_U=2 ...) through PFST_HIP_LIB.

  python tools/bn_bwd_apply_probe.py [tag]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfst_amd import hip_ops as H      # noqa: E402


def timeit(fn, reps=20):
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
    tag = sys.argv[1] if len(sys.argv) > 1 else 'lib'
    n = 8
    other = torch.empty(96 * 1024 * 1024, device='cuda')
    t_fill = timeit(lambda: other.fill_(1.0))
    tot_a = tot_w = 0.0
    for c, hw, relu_mask in ((2048, 128, True), (512, 128, False), (1024, 128, True), (256, 128, False), (256, 256, True), (64, 256, False), (128, 128, False)):
        x = torch.randn(n, c, hw, hw, device='cuda')
        dy = torch.randn_like(x)
        dx = torch.empty_like(x)
        g = torch.rand(c, device='cuda') + 0.5
        b = torch.randn(c, device='cuda') * 0.1
        mean, invstd = H.bn_stats(x)
        dg, db = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
        slots = 16
        part = torch.randn(2 * c * slots, device='cuda')
        mask = torch.randint(-2 ** 62, 2 ** 62, (x.numel() // 64,), device='cuda', dtype=torch.int64) if relu_mask else None

        def apply_only():
            other.fill_(1.0)
            H.bn_backward(dy, None, x, mean, invstd, g, dg, db, relu=True, dx=dx, beta=b, mask=mask, partials=part, slots=slots)

        def whole():
            other.fill_(1.0)
            H.bn_backward(dy, None, x, mean, invstd, g, dg, db, relu=True, dx=dx, beta=b, mask=mask)
        ta, tw = timeit(apply_only) - t_fill, timeit(whole) - t_fill
        gb = x.numel() * 4 / 1e9
        tot_a += ta
        tot_w += tw
        print(f'[{tag}] {c:5d} ch x {hw}^2 ({gb * 1e3:.0f} MB, gate from the {"bitmask" if relu_mask else "pre-BN tensor"}): apply {ta:.3f} ms = {3 * gb / ta:.2f} TB/s'
              f' | reduce + apply {tw:.3f} ms = {5 * gb / tw:.2f} TB/s', flush=True)
        del x, dy, dx, mask
    print(f'[{tag}] sum: apply {tot_a:.3f} ms, reduce + apply {tot_w:.3f} ms')


if __name__ == '__main__':
    main()
