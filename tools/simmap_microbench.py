"""PFGSTLoss similarity map at the BASELINE shape (b=8, 512 x 128 x 128 decoded features, dilation 2): GB/s on the 268 MB read."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfst_amd import hip_ops as ops

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); t.append(s.elapsed_time(e))
    return min(t)

for (b, c, h, d) in [(8, 512, 128, 2), (8, 512, 64, 1), (8, 2048, 128, 2)]:
    x = torch.randn(b, c, h, h, device='cuda')
    sim, norm = ops.sim_map(x, d)
    g = torch.randn_like(sim); dx = torch.empty_like(x)
    nbytes = x.numel() * 4
    tf = timeit(lambda: ops.sim_map(x, d))
    tb = timeit(lambda: ops.sim_map_bwd(x, sim, norm, g, d, out=dx))
    print(f'b{b} c{c} {h}x{h} d{d}: sim_map {tf*1e3:7.1f} us = {nbytes/tf/1e6:6.0f} GB/s (read {nbytes/1e6:.0f} MB);  '
          f'sim_map_bwd {tb*1e3:7.1f} us = {2*nbytes/tb/1e6:6.0f} GB/s (read + write)', flush=True)
