"""Winograd F(2x2,3x3) and F(4x4,3x3) vs the direct K-quad kernels at the BASELINE 3x3 layer shapes (b=8, 1024^2): ms per call
(direct / m=2 / m=4), the speed-up of each over direct, and the difference of the results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfst_amd import hip_ops as ops

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); t = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); t.append(s.elapsed_time(e))
    return min(t)

B = 8
for name, ci, co, d, hin in [('head.bottleneck', 2560, 512, 1, 128), ('l4.conv2', 512, 512, 4, 128), ('aux.conv', 1024, 256, 1, 128),
                             ('l3.conv2', 256, 256, 2, 128), ('l2.conv2', 128, 128, 1, 128), ('l1.conv2', 64, 64, 1, 256)]:
    x = torch.randn(B, ci, hin, hin, device='cuda'); w = torch.randn(co, ci, 3, 3, device='cuda') * 0.05
    wf, wd = ops.pack_weight(w)
    y = ops.conv_fprop(x, wf, co, 3, 1, d, d)
    dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.zeros_like(w); yw = torch.empty_like(y)
    r = {'fprop': [timeit(lambda: ops.conv_fprop(x, wf, co, 3, 1, d, d, out=y))],
         'dgrad': [timeit(lambda: ops.conv_dgrad(dy, wd, ci, (hin, hin), 3, 1, d, d, out=dx))],
         'wgrad': [timeit(lambda: ops.conv_wgrad_(dw, x, dy, 3, 1, d, d))]}
    errs = []
    for m in (2, 4):
        uf, ud = ops.wino_pack_weight(w, m=m)
        ops.wino_conv(x, uf, co, d, out=yw, m=m)
        errs.append(float((y - yw).norm() / y.norm()))
        r['fprop'].append(timeit(lambda: ops.wino_conv(x, uf, co, d, out=yw, m=m)))
        r['dgrad'].append(timeit(lambda: ops.wino_conv(dy, ud, ci, d, out=dx, m=m)))
        r['wgrad'].append(timeit(lambda: ops.wino_wgrad_(dw, x, dy, d, m=m)))
        del uf, ud
    print(f'{name:16s} diff {errs[0]:.1e} {errs[1]:.1e} | ' + ' | '.join(
        f'{k} {a:7.3f} {b:7.3f} {c:7.3f} ms  x{a / b:4.2f} x{a / c:4.2f}' for k, (a, b, c) in r.items()), flush=True)
    del x, w, y, yw, dy, dx, dw
    ops._wino_cache.clear(); torch.cuda.empty_cache()
