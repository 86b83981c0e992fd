"""Depthwise 3x3 kernels at the BASELINE layer shapes: time and achieved HBM rate (read + write of the plane, + the old values when accumulating)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfst_amd import hip_ops as ops


def timeit(fn, reps=7):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); t.append(s.elapsed_time(e))
    return sorted(t)[len(t) // 2]


for name, c, hw, dil in [('aspp d12', 2048, 128, 12), ('aspp d24', 2048, 128, 24), ('aspp d36', 2048, 128, 36), ('sep 560', 560, 256, 1), ('sep 512', 512, 256, 1)]:
    x = torch.randn(8, c, hw, hw, device='cuda'); w = torch.randn(c, 1, 3, 3, device='cuda'); y = torch.empty_like(x)
    dw = torch.zeros_like(w)
    nb = x.numel() * 4
    for label, fn, k in [('fprop', lambda: ops.dwconv(x, w, dil, out=y), 2), ('fprop+stats', lambda: ops.dwconv(x, w, dil, out=y, want_stats=True), 2),
                         ('dgrad acc', lambda: ops.dwconv(x, w, dil, flip=True, out=y, accumulate=True), 3), ('wgrad', lambda: ops.dwconv_wgrad_(dw, x, y, dil), 2)]:
        t = timeit(fn)
        print(f'{name:9s} {label:12s} {t:7.3f} ms  {k * nb / t / 1e9:5.2f} TB/s', flush=True)
