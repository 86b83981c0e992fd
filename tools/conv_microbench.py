"""Per-layer timing of the MFMA convolution kernels at the BASELINE shape (b=8, 1024^2): TFLOP/s for fprop, dgrad, wgrad."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfst_amd import hip_ops as ops

B = int(os.environ.get('B', 8)); S = int(os.environ.get('S', 1024))
s2, s4, s8 = S // 2, S // 4, S // 8
# (name, cin, cout, k, stride, dil, Hin, count per forward)
LAYERS = [
    ('stem.0', 3, 32, 3, 2, 1, S, 1), ('stem.3', 32, 32, 3, 1, 1, s2, 1), ('stem.6', 32, 64, 3, 1, 1, s2, 1),
    ('l1.conv1a', 64, 64, 1, 1, 1, s4, 1), ('l1.conv1', 256, 64, 1, 1, 1, s4, 2), ('l1.conv2', 64, 64, 3, 1, 1, s4, 3),
    ('l1.conv3', 64, 256, 1, 1, 1, s4, 4), ('l2.conv1a', 256, 128, 1, 1, 1, s4, 1), ('l2.conv2s', 128, 128, 3, 2, 1, s4, 1),
    ('l2.down', 256, 512, 1, 2, 1, s4, 1), ('l2.conv1', 512, 128, 1, 1, 1, s8, 3), ('l2.conv2', 128, 128, 3, 1, 1, s8, 3),
    ('l2.conv3', 128, 512, 1, 1, 1, s8, 4), ('l3.conv1a', 512, 256, 1, 1, 1, s8, 1), ('l3.conv2', 256, 256, 3, 1, 2, s8, 6),
    ('l3.conv3', 256, 1024, 1, 1, 1, s8, 6), ('l3.down', 512, 1024, 1, 1, 1, s8, 1), ('l3.conv1', 1024, 256, 1, 1, 1, s8, 5),
    ('l4.conv1a', 1024, 512, 1, 1, 1, s8, 1), ('l4.conv2', 512, 512, 3, 1, 4, s8, 3), ('l4.conv3', 512, 2048, 1, 1, 1, s8, 3),
    ('l4.down', 1024, 2048, 1, 1, 1, s8, 1), ('l4.conv1', 2048, 512, 1, 1, 1, s8, 2), ('aspp.pw', 2048, 512, 1, 1, 1, s8, 4),
    ('head.bottleneck', 2560, 512, 3, 1, 1, s8, 1), ('c1', 256, 48, 1, 1, 1, s4, 1), ('sep0.pw', 560, 512, 1, 1, 1, s4, 1),
    ('sep1.pw', 512, 512, 1, 1, 1, s4, 1), ('conv_seg', 512, 6, 1, 1, 1, s4, 1), ('aux.conv', 1024, 256, 3, 1, 1, s8, 1),
]
only = os.environ.get('ONLY')

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); t.append(s.elapsed_time(e))
    return min(t)

tot = {'fprop': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}; totf = 0.0
print(f'{"layer":16s} {"GFLOP":>8s} | fprop ms  TF/s | dgrad ms  TF/s | wgrad ms  TF/s | x count')
for name, ci, co, k, st, d, hin, cnt in LAYERS:
    if only and only not in name: continue
    pad = d if k == 3 else 0
    x = torch.randn(B, ci, hin, hin, device='cuda'); w = torch.randn(co, ci, k, k, device='cuda') * 0.05
    wf, wd = ops.pack_weight(w)
    y = ops.conv_fprop(x, wf, co, k, st, d, pad)
    dy = torch.randn_like(y); dw = torch.zeros_like(w); dx = torch.empty_like(x)
    fl = 2.0 * y.numel() * ci * k * k
    tf = timeit(lambda: ops.conv_fprop(x, wf, co, k, st, d, pad, out=y))
    td = timeit(lambda: ops.conv_dgrad(dy, wd, ci, (hin, hin), k, st, d, pad, out=dx))
    tw = timeit(lambda: ops.conv_wgrad_(dw, x, dy, k, st, d, pad))
    print(f'{name:16s} {fl/1e9:8.1f} | {tf:7.3f} {fl/tf/1e9:6.1f} | {td:7.3f} {fl/td/1e9:6.1f} | {tw:7.3f} {fl/tw/1e9:6.1f} | x{cnt}', flush=True)
    tot['fprop'] += tf * cnt; tot['dgrad'] += td * cnt; tot['wgrad'] += tw * cnt; totf += fl * cnt
    del x, w, y, dy, dw, dx
print('per forward-equivalent: GFLOP %.0f  fprop %.1f ms (%.1f TF/s)  dgrad %.1f ms (%.1f)  wgrad %.1f ms (%.1f)' % (
    totf / 1e9, tot['fprop'], totf / tot['fprop'] / 1e9, tot['dgrad'], totf / tot['dgrad'] / 1e9, tot['wgrad'], totf / tot['wgrad'] / 1e9))
