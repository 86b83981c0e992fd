"""Is the f16x3 GEMM bound by what it issues, or by the clock the chip holds under its matrix load (MI355X_MICROARCH.md, 'DVFS give-back')?

The same launches -- identical instruction streams, identical memory traffic -- on random operands, on all-zero activations and on all-zero
operands: cycles per MFMA do not depend on the data, the clock the chip sustains does (zero operands toggle almost nothing in the matrix
pipe).  If the zero-data run is much faster, the random-data rate is set by power / clock, and removing vector instructions from the loop
cannot help (round 4: one third fewer split instructions changed the step by 0.0 %, profiles/r04_ab_split4_vs_split6.txt), nor does a
timing build that feeds constants instead of split activations measure the value of pre-split operands (round 3's 441 vs 349 TFLOP/s-eq
"no-split" diagnostic).

    python tools/dvfs_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfst_amd import hip_ops as H      # noqa: E402


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n, hw = 8, 128
    print('1x1 convolution forward, b = 8 x 128^2 pixels, f16x3 (three fp16 MFMAs per product); TFLOP/s-equivalent = algorithmic flops / time')
    for ci, co in ((2048, 512), (512, 2048), (1024, 256), (256, 1024)):
        fl = 2.0 * n * ci * co * hw * hw
        out = torch.empty(n, co, hw, hw, device='cuda')
        row = []
        for xname, wname in (('random', 'random'), ('zeros', 'random'), ('random', 'zeros'), ('zeros', 'zeros')):
            x = torch.relu(torch.randn(n, ci, hw, hw, device='cuda')) if xname == 'random' else torch.zeros(n, ci, hw, hw, device='cuda')
            w = torch.randn(co, ci, 1, 1, device='cuda') * (2.0 / ci) ** 0.5 if wname == 'random' else torch.zeros(co, ci, 1, 1, device='cuda')
            w4f, _, wa = H.pack_weight_f16x2(w, True, False)
            xa = H.absmax(x)
            # keep the chip under load for a while before timing: the clock settles over hundreds of milliseconds
            for _ in range(200):
                H.conv_fprop_f16x3(x, w4f, wa, xa, co, 1, out=out)
            t = timeit(lambda: H.conv_fprop_f16x3(x, w4f, wa, xa, co, 1, out=out), reps=100)
            row.append((xname, wname, t, fl / t / 1e9))
        base = row[0][2]
        print(f'  K={ci:4d} M={co:4d}: ' + ' | '.join(f'x {a:6s} w {b:6s} {t:.3f} ms {tf:4.0f} TF-eq ({100 * (base / t - 1):+.0f} %)' for a, b, t, tf in row), flush=True)


if __name__ == '__main__':
    main()
