#!/usr/bin/env python3
"""Launch-level times of the short kernels that sit between the step's large launches (round 5): the statistics finalisers of the fused
BatchNorm partials (forward and backward) at the slot counts the workload produces, and the fused up-sample + cross-entropy forward.
Run once per library build (PFST_HIP_LIB=...) in the same gpurun call for an A/B.

  python tools/small_kernel_probe.py [tag]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def timeit(fn, n=200, flush=None):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(n):
        if flush is not None:
            flush.add_(1.0)                 # the producer's traffic: the slots do not sit in the cache of the CU that reads them
        a.record()
        fn()
        b.record()
        b.synchronize()
        tot += a.elapsed_time(b)
    return tot / n * 1e3


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'lib'
    from pfst_amd import hip_ops as H
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    flush = torch.zeros(64 << 20, device=dev)
    print(f'[{tag}] us per launch (events around one launch; 256 MB written between launches)')
    # (channels, slots): 1x1 / Winograd launches at 1/8 resolution (b=8: 8 * 128 * 128 / 128 px * 2 = 2048 slots), layer1 (8192), the stem (32768)
    for c, t in ((2048, 2048), (512, 2048), (256, 2048), (1024, 2048), (256, 8192), (64, 8192), (64, 32768), (128, 2048)):
        st = torch.randn(4 * c * t, device=dev)
        g, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        am = H.amax_slots(dev)
        t0 = timeit(lambda: H.bn_finalize_partials(st, t, c, 8.0 * 128 * 128, rm, rv, gamma=g, beta=b), flush=flush)
        t1 = timeit(lambda: H.bn_finalize_partials(st, t, c, 8.0 * 128 * 128, rm, rv, gamma=g, beta=b, predict_amax=am), flush=flush)
        print(f'  bn_finalize_partials C={c:5d} T={t:6d}: sums {t0:7.1f}   sums + (min, max) {t1:7.1f}')
    for hw, S in ((256, 1024), (128, 1024), (128, 512)):
        lg = torch.randn(8, 6, hw, hw, device=dev)
        lab = torch.randint(0, 6, (8, S, S), device=dev, dtype=torch.uint8)
        pw = torch.rand(8, S, S, device=dev)
        t0 = timeit(lambda: H.ce_upsample_fwd(lg, lab), n=50)
        t1 = timeit(lambda: H.ce_upsample_fwd(lg, lab, pix_weight=pw), n=50)
        lse, acc = H.ce_upsample_fwd(lg, lab, pix_weight=pw)
        lab[:, :7] = 255
        out = torch.empty_like(lg)
        t2 = timeit(lambda: H.ce_upsample_bwd(lg, lab, lse, 1e-3, pix_weight=pw, out=out), n=50)
        print(f'  ce_upsample_bwd 8 x 6 x {hw}^2 <- {S}^2: {t2:7.1f}')
        print(f'  ce_upsample_fwd 8 x 6 x {hw}^2 -> {S}^2: {t0:7.1f}   with pixel weights {t1:7.1f}   (incl. two small allocations)  acc {acc.tolist()}')


if __name__ == '__main__':
    main()
