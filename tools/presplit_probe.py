"""What is a PRE-SPLIT activation operand worth to the f16x3 GEMMs?

The Winograd-domain GEMMs and weight-gradient products already exist in both forms on identical shapes: V / dM written plain by the
transforms and split in registers inside the GEMM loop (scale from the measured max |V|), or written packed (one dword = two fp16 pieces)
and only byte-permuted in the loop.  A 1x1 convolution is the same kernel (ONE variant) on one filter set, so timing the two forms of the
36-set launch at the network's (K, M) pairs prices what BatchNorm kernels writing packed activations would buy the 1x1 layers -- before
any producer is taught to bound max |y| ahead of writing y.

    python tools/presplit_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfst_amd import hip_ops as ops          # noqa: E402
from pfst_amd._lib import call               # noqa: E402


def timeit(fn, reps=10):
    for _ in range(3):
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
    dev = torch.device('cuda')
    st = torch.cuda.current_stream().cuda_stream
    m, nx, n, hw = 4, 36, 8, 128
    print('Winograd-domain GEMM (36 sets x 8 images x 1024 tiles = the pixel count of a 1x1 layer at 1/8 resolution x 2.25), in-register split vs pre-split:')
    for ci, co in ((512, 512), (2560, 512), (256, 256), (1024, 256), (128, 128), (512, 2560)):
        x = torch.relu(torch.randn(n, ci, hw, hw, device=dev))
        w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        dy = torch.randn(n, co, hw, hw, device=dev) * 1e-3
        t = ops.wino_tiles(hw, hw, 1, m)
        uf, _, af, _ = ops.wino_pack_weight_f16(w, True, False, m=m)
        xa, da = ops.absmax(x), ops.absmax(dy)
        mb = torch.empty(nx * n * co * t, device=dev)
        res = {}
        for packed in (0, 1):
            v = torch.empty(nx * n * ci * t, device=dev)
            dm = torch.empty(nx * n * co * t, device=dev)
            va, dma = ops.amax_slots(dev), ops.amax_slots(dev)
            call('pfst_wino_input', x.data_ptr(), ci * hw * hw, v.data_ptr(), n, ci, hw, hw, 1, m, va.data_ptr(), xa.data_ptr() if packed else 0, 0, st)
            call('pfst_wino_dy', dy.data_ptr(), co * hw * hw, dm.data_ptr(), n, co, hw, hw, 1, m, dma.data_ptr(), da.data_ptr() if packed else 0, st)
            tg = timeit(lambda: call('pfst_wino_gemm_f16x3', v.data_ptr(), uf.data_ptr(), af.data_ptr(), va.data_ptr(), mb.data_ptr(), n, ci, co, t, m, packed, st))
            du = torch.empty(nx * co * ci, device=dev)
            dw = torch.zeros(co, ci, 3, 3, device=dev)
            tw = timeit(lambda: call('pfst_wino_wgrad', v.data_ptr(), dm.data_ptr(), du.data_ptr(), dw.data_ptr(), n, ci, co, t, m, 2, va.data_ptr(),
                                     dma.data_ptr(), packed, st)) if co > 64 else float('nan')
            res[packed] = (tg, tw)
            del v, dm
        fl = 2.0 * nx * n * ci * co * t
        (g0, w0), (g1, w1) = res[0], res[1]
        print(f'  K={ci:5d} M={co:5d}: GEMM split-in-loop {g0:.3f} ms ({fl / g0 / 1e9:4.0f} TF-eq)  pre-split {g1:.3f} ms ({fl / g1 / 1e9:4.0f})  {100 * (g1 / g0 - 1):+.1f} % | '
              f'wgrad {w0:.3f} -> {w1:.3f} ms  {100 * (w1 / w0 - 1):+.1f} %', flush=True)
        del x, w, dy, mb
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
