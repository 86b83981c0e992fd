#!/usr/bin/env python3
"""The fused depthwise backward (dwconv3x3_kernel<3, true> with the BatchNorm-backward fold) run ALONE, beside a rocBLAS GEMM and beside the
f16x3 weight-gradient kernel on a second stream, against a quiet reference and against fp64 autograd.  Round 5 found the build with packed-fp32
code (clang's SLP vectoriser: v_pk_fma_f32 / v_pk_add_f32 on pairs of weight-gradient accumulators) returning WRONG sums for one accumulator
(tap 6) of a few channels whenever the weight-gradient kernel shared the CUs -- the case the product's stream overlap creates; the build
without packed code (pfst_amd/build.py: -fno-slp-vectorize for the streaming kernels) is exact.   python tools/race_probe.py
(PFST_HIP_LIB=<variant>: python -m pfst_amd.build --variant slp  builds the default flags for every file under ab_libs/)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from pfst_amd import hip_ops as ops  # noqa: E402


def main():
    torch.manual_seed(0)
    n, c, h, w = 2, 560, 32, 32
    x = torch.randn(n, c, h, w, device='cuda'); dy = torch.randn(n, c, h, w, device='cuda'); wt = torch.randn(c, 1, 3, 3, device='cuda')
    gamma = torch.rand(c, device='cuda') + 0.5; beta = torch.randn(c, device='cuda') * 0.1
    pre = ops.dwconv(x, wt, 1)
    mean, invstd, coef = ops.bn_stats(pre, gamma=gamma, beta=beta)
    side = torch.cuda.Stream()
    X = torch.randn(2, 512, 64, 64, device='cuda'); DY = torch.randn(2, 512, 64, 64, device='cuda'); DW = torch.zeros(512 * 512, device='cuda')
    xa, dya = ops.absmax(X), ops.absmax(DY)
    dg0 = torch.zeros(c, device='cuda'); db0 = torch.zeros(c, device='cuda')
    rec = ops.bn_backward_sums(dy, pre, mean, invstd, gamma, beta, dg0, db0)
    torch.cuda.synchronize()
    ref = torch.zeros(c * 9, device='cuda'); dxr = torch.empty_like(x)
    ops.dwconv_bwd_(ref, x, dy, wt, 1, dxr, bnb=(pre, rec)); torch.cuda.synchronize()
    # truth: dL/dpre by the two-pass BatchNorm backward kernel, then fp64 autograd of the depthwise convolution
    dgx = torch.zeros(c, device='cuda'); dbx = torch.zeros(c, device='cuda')
    dpre = ops.bn_backward(dy, None, pre, mean, invstd, gamma, dgx, dbx, True, None, False, beta=beta)
    w64 = wt.double().clone().requires_grad_(True)
    F.conv2d(x.double(), w64, None, 1, 1, 1, c).backward(dpre.double())
    truth = w64.grad.reshape(-1)
    print('quiet run vs fp64 autograd, worst per tap:', [f'{float(v):.1e}' for v in (ref.double() - truth).view(c, 9).abs().max(dim=0)[0]])
    for busy in ('none', 'rocBLAS GEMM', 'f16x3 weight gradient'):
        bad, worst = 0, 0.0
        for rep in range(8):
            dw = torch.zeros(c * 9, device='cuda'); dx = torch.empty_like(x)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(4):
                    if busy == 'f16x3 weight gradient':
                        ops.conv_wgrad_f16x3_(DW, X, DY, xa, dya)
                    elif busy == 'rocBLAS GEMM':
                        X.view(2, 512, -1) @ DY.view(2, 512, -1).transpose(1, 2)
            ops.dwconv_bwd_(dw, x, dy, wt, 1, dx, bnb=(pre, rec))
            torch.cuda.synchronize()
            bad += int(not torch.equal(dw, ref))
            worst = max(worst, float((dw.double() - truth).abs().max()))
            assert torch.equal(dx, dxr)
        print(f'beside {busy:22s}: {bad} of 8 runs differ from the quiet run; worst |dw - truth| {worst:.2e}')


if __name__ == '__main__':
    main()
