"""bf16x6 split conv vs fp32-MFMA conv: accuracy against an fp64 reference and speed at BASELINE layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pfst_amd import hip_ops as ops

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); t = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); t.append(s.elapsed_time(e))
    return min(t)

# accuracy on a small case vs fp64
g = torch.Generator().manual_seed(0)
x = torch.randn(2, 64, 20, 24, generator=g); w = torch.randn(96, 64, 3, 3, generator=g) * 0.1
ref = F.conv2d(x.double(), w.double(), None, 1, 2, 2)
xd, wd = x.cuda(), w.cuda()
wf, wdg = ops.pack_weight(wd); w6f, w6d = ops.pack_weight_split(wd)
y32 = ops.conv_fprop(xd, wf, 96, 3, 1, 2, 2).cpu().double()
y6 = ops.conv_fprop_split(xd, w6f, 96, 3, 1, 2, 2).cpu().double()
print('fprop rel err vs fp64: fp32-MFMA %.2e  bf16x6 %.2e' % (float((y32 - ref).norm() / ref.norm()), float((y6 - ref).norm() / ref.norm())))
dy = torch.randn(ref.shape, generator=g)
dxr = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), 1, 2, 2)
d32 = ops.conv_dgrad(dy.cuda(), wdg, 64, (20, 24), 3, 1, 2, 2).cpu().double()
d6 = ops.conv_dgrad_split(dy.cuda(), w6d, 64, (20, 24), 3, 1, 2, 2).cpu().double()
print('dgrad rel err vs fp64: fp32-MFMA %.2e  bf16x6 %.2e' % (float((d32 - dxr).norm() / dxr.norm()), float((d6 - dxr).norm() / dxr.norm())))
B = 8
for name, ci, co, k, st, d, hin in [('l4.conv2', 512, 512, 3, 1, 4, 128), ('aspp.pw', 2048, 512, 1, 1, 1, 128), ('head.bottleneck', 2560, 512, 3, 1, 1, 128),
                                   ('l3.conv3', 256, 1024, 1, 1, 1, 128), ('l1.conv2', 64, 64, 3, 1, 1, 256), ('sep1.pw', 512, 512, 1, 1, 1, 256)]:
    pad = d if k == 3 else 0
    x = torch.randn(B, ci, hin, hin, device='cuda'); w = torch.randn(co, ci, k, k, device='cuda') * 0.05
    wf, wdg = ops.pack_weight(w); w6f, w6d = ops.pack_weight_split(w)
    y = ops.conv_fprop(x, wf, co, k, st, d, pad); y6 = ops.conv_fprop_split(x, w6f, co, k, st, d, pad)
    fl = 2.0 * y.numel() * ci * k * k
    t32 = timeit(lambda: ops.conv_fprop(x, wf, co, k, st, d, pad, out=y)); t6 = timeit(lambda: ops.conv_fprop_split(x, w6f, co, k, st, d, pad, out=y6))
    err = float((y6 - y).norm() / y.norm())
    print(f'{name:16s} fp32-MFMA {t32:7.3f} ms {fl/t32/1e9:6.1f} TF/s | bf16x6 {t6:7.3f} ms {fl/t6/1e9:6.1f} TF/s-equiv ({6*fl/t6/1e9:6.0f} bf16 TF/s) | diff {err:.1e}', flush=True)
print('--- wgrad')
for name, ci, co, k, st, d, hin in [('l4.conv2', 512, 512, 3, 1, 4, 128), ('aspp.pw', 2048, 512, 1, 1, 1, 128), ('head.bottleneck', 2560, 512, 3, 1, 1, 128),
                                   ('l3.conv3', 256, 1024, 1, 1, 1, 128), ('l1.conv2', 64, 64, 3, 1, 1, 256), ('sep1.pw', 512, 512, 1, 1, 1, 256)]:
    pad = d if k == 3 else 0
    x = torch.randn(B, ci, hin, hin, device='cuda'); dy = torch.randn(B, co, hin, hin, device='cuda')
    dw = torch.zeros(co, ci, k, k, device='cuda'); dw6 = torch.zeros_like(dw)
    fl = 2.0 * dy.numel() * ci * k * k
    t32 = timeit(lambda: ops.conv_wgrad_(dw, x, dy, k, st, d, pad)); t6 = timeit(lambda: ops.conv_wgrad_split_(dw6, x, dy, k, st, d, pad))
    print(f'{name:16s} fp32-MFMA {t32:7.3f} ms {fl/t32/1e9:6.1f} TF/s | bf16x6 {t6:7.3f} ms {fl/t6/1e9:6.1f} TF/s-equiv', flush=True)
