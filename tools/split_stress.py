"""Race hunt for the pipelined bf16x6 kernels: many launches over random shapes, each compared with the fp32-MFMA kernel (a missed barrier
shows up as a sporadic large error, not as a tolerance-sized one).  Exit code 1 on any mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import random
import torch
from pfst_amd import hip_ops as ops



def run(iters, seed=0):
    random.seed(seed)
    torch.manual_seed(seed)
    bad = 0
    n_launch = 0
    for it in range(iters):
        bad, n_launch = _one(it, bad, n_launch)
    torch.cuda.synchronize()
    return n_launch, bad


def _one(it, bad, n_launch):
    if True:
        k = random.choice([1, 1, 3])
        ci = random.choice([32, 64, 96, 128, 256, 512, 560, 1024]) if k == 1 else random.choice([64, 96, 128, 144, 256])
        if ci * k * k < 512:
            ci = 512 if k == 1 else 64
        co = random.choice([80, 128, 192, 256, 320, 512])
        h, w = random.randint(9, 40), random.randint(9, 48)
        if it % 8 == 0:                      # full machine: every CU holds its 2-3 workgroups
            h, w = random.choice([96, 112, 128]), 128
        dil = 1 if k == 1 else random.choice([1, 2, 4])
        pad = dil if k == 3 else 0
        n = random.randint(1, 3)
        x = torch.randn(n, ci, h, w, device='cuda'); wt = torch.randn(co, ci, k, k, device='cuda') * 0.05
        wf, wd = ops.pack_weight(wt); w6f, w6d = ops.pack_weight_split(wt)
        y32 = ops.conv_fprop(x, wf, co, k, 1, dil, pad)
        dy = torch.randn_like(y32)
        d32 = ops.conv_dgrad(dy, wd, ci, (h, w), k, 1, dil, pad)
        for rep in range(4):
            y6 = ops.conv_fprop_split(x, w6f, co, k, 1, dil, pad)
            d6 = ops.conv_dgrad_split(dy, w6d, ci, (h, w), k, 1, dil, pad)
            n_launch += 2
            for name, a, b in (('fprop', y6, y32), ('dgrad', d6, d32)):
                err = float((a - b).abs().max() / b.abs().max())
                if not err < 2e-5:
                    bad += 1
                    print(f'MISMATCH {name} it {it} rep {rep} shape n{n} ci{ci} co{co} {h}x{w} k{k} d{dil}: max rel {err:.3e}', flush=True)
        if k == 1 and (h * w) % 4 == 0:
            dw32 = torch.zeros_like(wt); ops.conv_wgrad_(dw32, x, dy, k, 1, dil, pad)
            for rep in range(3):
                dw6 = torch.zeros_like(wt); ops.conv_wgrad_split_(dw6, x, dy, k, 1, dil, pad)
                n_launch += 1
                err = float((dw6 - dw32).abs().max() / dw32.abs().max())
                if not err < 5e-5:
                    bad += 1
                    print(f'MISMATCH wgrad it {it} rep {rep} n{n} ci{ci} co{co} {h}x{w}: max rel {err:.3e}', flush=True)
    return bad, n_launch


if __name__ == '__main__':
    n_launch, bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 120)
    print(f'{n_launch} split launches, {bad} mismatches')
    sys.exit(1 if bad else 0)
