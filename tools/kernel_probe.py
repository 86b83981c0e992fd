"""A few launches of chosen conv kernels at BASELINE layer shapes, for `rocprofv3 --pmc ...` counter passes.
usage: python3 tools/kernel_probe.py [fprop|dgrad|wgrad|split|wsplit ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfst_amd import hip_ops as ops

which = sys.argv[1:] or ['fprop', 'wgrad']
B = 8
LAYERS = [('l4.conv2', 512, 512, 3, 1, 4, 128), ('l4.down', 1024, 2048, 1, 1, 1, 128)]
for name, ci, co, k, st, d, hin in LAYERS:
    pad = d if k == 3 else 0
    x = torch.randn(B, ci, hin, hin, device='cuda'); w = torch.randn(co, ci, k, k, device='cuda') * 0.05
    wf, wd = ops.pack_weight(w)
    y = ops.conv_fprop(x, wf, co, k, st, d, pad)
    dy = torch.randn_like(y); dw = torch.zeros_like(w); dx = torch.empty_like(x)
    for _ in range(3):
        if 'fprop' in which: ops.conv_fprop(x, wf, co, k, st, d, pad, out=y)
        if 'dgrad' in which: ops.conv_dgrad(dy, wd, ci, (hin, hin), k, st, d, pad, out=dx)
        if 'wgrad' in which: ops.conv_wgrad_(dw, x, dy, k, st, d, pad)
        if 'split' in which:
            w6f, w6d = ops.pack_weight_split(w)
            ops.conv_fprop_split(x, w6f, co, k, st, d, pad, out=y)
        if 'wsplit' in which: ops.conv_wgrad_split_(dw, x, dy, k, st, d, pad)
    torch.cuda.synchronize()
print('done')
