"""Achieved HBM GB/s of the memory-bound kernels at BASELINE shapes (b=8, 1024^2 tiles): algorithmic bytes / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfst_amd import hip_ops as ops

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); t = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); t.append(s.elapsed_time(e))
    return min(t)

B = 8
rows = []
for name, C, H, d in [('aspp.dw d12', 2048, 128, 12), ('aspp.dw d24', 2048, 128, 24), ('aspp.dw d36', 2048, 128, 36),
                      ('sep0.dw d1', 560, 256, 1), ('sep1.dw d1', 512, 256, 1)]:
    x = torch.randn(B, C, H, H, device='cuda'); w = torch.randn(C, 1, 3, 3, device='cuda'); y = torch.empty_like(x); dw = torch.zeros_like(w)
    nb = x.numel() * 4
    t = timeit(lambda: ops.dwconv(x, w, d, out=y)); rows.append((name + ' fwd', 2 * nb, t))
    t = timeit(lambda: ops.dwconv(x, w, d, flip=True, out=y)); rows.append((name + ' dgrad', 2 * nb, t))
    t = timeit(lambda: ops.dwconv_wgrad_(dw, x, y, d)); rows.append((name + ' wgrad', 2 * nb, t))
    del x, y
for name, C, H in [('bn 256ch@256', 256, 256), ('bn 2048ch@128', 2048, 128), ('bn 64ch@512', 64, 512)]:
    x = torch.randn(B, C, H, H, device='cuda'); g = torch.rand(C, device='cuda') + .5; b = torch.randn(C, device='cuda')
    nb = x.numel() * 4
    mean, invstd = ops.bn_stats(x)
    t = timeit(lambda: ops.bn_stats(x)); rows.append((name + ' stats', nb, t))
    y = torch.empty_like(x)
    t = timeit(lambda: ops.bn_apply(x, mean, invstd, g, b, True, out=y)); rows.append((name + ' apply', 2 * nb, t))
    t = timeit(lambda: ops.bn_apply(x, mean, invstd, g, b, True, residual=x, out=y)); rows.append((name + ' apply+res', 3 * nb, t))
    dy = torch.randn_like(x); dg = torch.zeros(C, device='cuda'); db = torch.zeros(C, device='cuda'); dx = torch.empty_like(x)
    t = timeit(lambda: ops.bn_backward(dy, y, x, mean, invstd, g, dg, db, True, dx=dx)); rows.append((name + ' bwd', 7 * nb, t))
    del x, y, dy, dx
print(f'{"kernel":28s} {"MB":>9s} {"ms":>8s} {"GB/s":>8s} {"of 8 TB/s":>9s}')
for name, nb, t in rows:
    print(f'{name:28s} {nb/1e6:9.1f} {t:8.3f} {nb/t/1e6:8.0f} {nb/t/1e6/8000:9.2f}')
