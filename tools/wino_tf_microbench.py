"""HBM efficiency of the Winograd transform kernels per layer shape: GB/s on the algorithmic bytes (input/dY transform: read N,
write (m+2)^2/m^2 N; output transform: read (m+2)^2/m^2 N, write N)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfst_amd import hip_ops as ops
from pfst_amd._lib import call

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); t = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); t.append(s.elapsed_time(e))
    return min(t)

B = 8
st = torch.cuda.current_stream().cuda_stream
for m in (4, 2):
    nx = (m + 2) ** 2
    for name, c, d, hw in [('head 2560 d1', 2560, 1, 128), ('l4 512 d4', 512, 4, 128), ('l4 512 d2', 512, 2, 128), ('l3 256 d2', 256, 2, 128),
                           ('aux 1024 d1', 1024, 1, 128), ('l2 128 d1', 128, 1, 128)]:
        x = torch.randn(B, c, hw, hw, device='cuda'); y = torch.empty_like(x)
        t = ops.wino_tiles(hw, hw, d, m)
        v = torch.empty(nx * B * c * t, device='cuda')
        nb = x.numel() * 4
        ti = timeit(lambda: call('pfst_wino_input', x.data_ptr(), c * hw * hw, v.data_ptr(), B, c, hw, hw, d, m, 0, 0, 0, st))
        to = timeit(lambda: call('pfst_wino_output', v.data_ptr(), y.data_ptr(), c * hw * hw, B, c, hw, hw, d, 0, 0, 0, 0, 0, 0, 0, m, st))
        td = timeit(lambda: call('pfst_wino_dy', x.data_ptr(), c * hw * hw, v.data_ptr(), B, c, hw, hw, d, m, 0, 0, st))
        f = 1 + nx / m ** 2
        print(f'm={m} {name:14s} input {ti:6.3f} ms {nb * f / ti / 1e6:6.0f} GB/s | output {to:6.3f} ms {nb * f / to / 1e6:6.0f} GB/s | dy {td:6.3f} ms {nb * f / td / 1e6:6.0f} GB/s', flush=True)
        del x, y, v
        torch.cuda.empty_cache()
