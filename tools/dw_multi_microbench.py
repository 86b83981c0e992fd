"""usage (GPU box): python tools/dw_multi_microbench.py        (round 4 compared the 512- and 1024-thread forms of the backward kernel with it: profiles/r04_dw_multi_microbench.txt; the 512-thread form was removed in round 5)
The ASPP head's fused depthwise launches at the bench shape (b = 8, 2048 channels, 128 x 128 planes, dilations 12 / 24 / 36): time per launch
and algorithmic GB/s (forward: x + 3 y; backward with BatchNorm backward folded in: x + 3 (dy + pre) + dx written)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import pfst_amd  # noqa: E402,F401
from pfst_amd import hip_ops as ops  # noqa: E402


def timed(fn, reps=10):
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
    dev = torch.device('cuda')
    n, c, h, w = 8, 2048, 128, 128
    dils = [12, 24, 36]
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(n, c, h, w, device=dev, generator=g)
    ws = [torch.randn(c, 1, 3, 3, device=dev, generator=g) for _ in dils]
    plane = 4.0 * n * c * h * w
    t = timed(lambda: ops.dwconv_multi(x, ws, dils, want_stats=True, want_mean=True))
    print(f'multi_fwd  {t:7.3f} ms  {4 * plane / t / 1e6:7.0f} GB/s')
    ys = [torch.randn(n, c, h, w, device=dev, generator=g) for _ in dils]          # the branches' pre-normalisation outputs
    dys = [torch.randn(n, c, h, w, device=dev, generator=g) for _ in dils]
    dws = [torch.zeros(c, 1, 3, 3, device=dev) for _ in dils]
    dx = torch.empty_like(x)
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    recs = []
    for y, dy in zip(ys, dys):
        mean, invstd, _ = ops.bn_stats(y, gamma=gamma, beta=beta)
        recs.append(ops.bn_backward_sums(dy, y, mean, invstd, gamma, beta, torch.zeros(c, device=dev), torch.zeros(c, device=dev)))
    mg = torch.randn(n, c, device=dev, generator=g)
    for label, kw, passes in (('multi_bwd (plain)        ', {}, 5), ('multi_bwd (+BN backward) ', {'bnb': list(zip(ys, recs))}, 8)):
        t = timed(lambda: ops.dwconv_multi_bwd_(dws, x, dys, ws, dils, dx, mean_grad=mg, **kw))
        print(f'{label} NT=1024 {t:7.3f} ms  {passes * plane / t / 1e6:7.0f} GB/s')


if __name__ == '__main__':
    main()
