#!/usr/bin/env python3
"""BatchNorm backward (reduce pass + apply pass, 5 N of traffic) on a tensor far larger than the 256 MiB Infinity Cache: whole tensor per pass, or
channel group by channel group (reduce then apply of a group whose dy + x + dx fit the cache, so that the apply pass re-reads from it)?

  python tools/bn_group_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfst_amd import hip_ops as H      # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
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
    n = 8
    for c, hw in ((1024, 128), (2048, 128), (256, 256), (512, 128)):
        x = torch.randn(n, c, hw, hw, device='cuda')
        dy = torch.randn_like(x)
        dx = torch.empty_like(x)
        g = torch.rand(c, device='cuda') + 0.5
        b = torch.randn(c, device='cuda') * 0.1
        mean, invstd = H.bn_stats(x)
        dg, db = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
        other = torch.empty(96 * 1024 * 1024, device='cuda')           # 384 MB written between repetitions: every repetition starts from a cold cache

        def whole():
            other.fill_(1.0)
            H.bn_backward(dy, None, x, mean, invstd, g, dg, db, relu=True, dx=dx, beta=b)

        def grouped(k):
            def f():
                other.fill_(1.0)
                step = c // k
                for c0 in range(0, c, step):
                    sl = slice(c0, c0 + step)
                    H.bn_backward(dy[:, sl], None, x[:, sl], mean[sl], invstd[sl], g[sl], dg[sl], db[sl], relu=True, dx=dx[:, sl], beta=b[sl])
            return f
        t_fill = timeit(lambda: other.fill_(1.0))
        tw = timeit(whole) - t_fill
        line = f'{c:5d} ch x {hw}^2 ({x.numel() * 4 / 1e6:.0f} MB per tensor): whole {tw:.3f} ms'
        for k in (2, 4, 8, 16):
            mb = 3 * x.numel() * 4 / k / 1e6
            line += f' | {k} groups ({mb:.0f} MB) {timeit(grouped(k)) - t_fill:.3f}'
        print(line, flush=True)
        del x, dy, dx, other


if __name__ == '__main__':
    main()
