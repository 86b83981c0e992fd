"""Stage-1 check of the two-piece fp16 split GEMM (csrc/conv_f16x3.hip): accuracy against fp64 next to the fp32-input MFMA and bf16x6
kernels, and throughput on the train step's large shapes.  python tools/f16x3_probe.py [--bench-only]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfst_amd import hip_ops as H      # noqa: E402


def rel(a, ref):
    return float((a.double() - ref).norm() / ref.norm())


def accuracy():
    import torch.nn.functional as F
    for (n, ci, co, hw, k, stride, dil, scale_x, scale_w) in [(2, 64, 128, 24, 1, 1, 1, 1.0, 1.0), (2, 96, 160, 20, 3, 1, 2, 1.0, 1.0),
                                                              (1, 128, 128, 18, 3, 2, 1, 1e-6, 30.0), (2, 512, 192, 16, 1, 1, 1, 3e4, 1e-3),
                                                              (1, 32, 72, 17, 3, 1, 1, 1.0, 1.0)]:
        g = torch.Generator().manual_seed(ci + co)
        pad = dil * (k // 2)
        x = torch.randn(n, ci, hw, hw, generator=g)
        x = (torch.relu(x) * (1 + x.abs()) * scale_x * torch.exp(2.0 * torch.randn(n, ci, hw, hw, generator=g))).cuda()   # wide dynamic range
        w = (torch.randn(co, ci, k, k, generator=g) * scale_w * (2.0 / (ci * k * k)) ** 0.5).cuda()
        ref = F.conv2d(x.double(), w.double(), None, stride, pad, dil)
        ho = ref.shape[-1]
        wf, wd = H.pack_weight(w, True, True)
        y32 = H.conv_fprop(x, wf, co, k, stride, dil, pad)
        w6f, w6d = H.pack_weight_split(w, True, co % 16 == 0)
        y6 = H.conv_fprop_split(x, w6f, co, k, stride, dil, pad)
        w4f, w4d, wa = H.pack_weight_f16x2(w, True, ci > 64 and co % 32 == 0)
        xa = H.absmax(x)
        y3 = H.conv_fprop_f16x3(x, w4f, wa, xa, co, k, stride, dil, pad)
        msg = f'{n}x{ci}->{co}@{hw} k{k} s{stride} d{dil}: fprop fp32 {rel(y32, ref):.2e} bf16x6 {rel(y6, ref):.2e} f16x3 {rel(y3, ref):.2e}'
        if w4d is not None:
            dy = (torch.randn(n, co, ho, ho, generator=g) * 1e-5 * torch.exp(1.5 * torch.randn(n, co, ho, ho, generator=g))).cuda()
            dref = torch.autograd.grad(F.conv2d(x.double().requires_grad_(), w.double(), None, stride, pad, dil), [], None, allow_unused=True) if False else None
            xd = x.double().requires_grad_()
            F.conv2d(xd, w.double(), None, stride, pad, dil).backward(dy.double())
            d32 = H.conv_dgrad(dy, wd, ci, (hw, hw), k, stride, dil, pad)
            d3 = H.conv_dgrad_f16x3(dy, w4d, wa, H.absmax(dy), ci, (hw, hw), k, stride, dil, pad)
            msg += f' | dgrad fp32 {rel(d32, xd.grad):.2e} f16x3 {rel(d3, xd.grad):.2e}'
        print(msg, flush=True)


def bench():
    def timeit(fn, reps=20):
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
    for (n, ci, co, hw, k, dil) in [(8, 2048, 512, 128, 1, 1), (8, 512, 2048, 128, 1, 1), (8, 1024, 256, 128, 1, 1), (8, 256, 1024, 128, 1, 1),
                                    (8, 512, 512, 128, 3, 2), (8, 512, 512, 256, 1, 1)]:
        pad = dil * (k // 2)
        x = torch.relu(torch.randn(n, ci, hw, hw, device='cuda'))
        w = torch.randn(co, ci, k, k, device='cuda') * (2.0 / (ci * k * k)) ** 0.5
        out = torch.empty(n, co, hw, hw, device='cuda')
        wf, _ = H.pack_weight(w, True, False)
        w6f, _ = H.pack_weight_split(w, True, False)
        w4f, _, wa = H.pack_weight_f16x2(w, True, False)
        xa = H.absmax(x)
        fl = 2.0 * n * ci * co * k * k * hw * hw
        t32 = timeit(lambda: H.conv_fprop(x, wf, co, k, 1, dil, pad, out=out))
        t6 = timeit(lambda: H.conv_fprop_split(x, w6f, co, k, 1, dil, pad, out=out))
        t3 = timeit(lambda: H.conv_fprop_f16x3(x, w4f, wa, xa, co, k, 1, dil, pad, out=out))
        ta = timeit(lambda: H.absmax(x, out=xa))
        print(f'{n}x{ci}->{co}@{hw} k{k} d{dil}: fp32 {t32:.3f} ms ({fl / t32 / 1e9:5.0f} TF)  bf16x6 {t6:.3f} ms ({fl / t6 / 1e9:5.0f})  '
              f'f16x3 {t3:.3f} ms ({fl / t3 / 1e9:5.0f} TF-eq, {3 * fl / t3 / 1e12:.2f} PF f16)   absmax(x) {ta:.3f} ms', flush=True)


def wgrad():
    def timeit(fn, reps=20):
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
    for (n, ci, co, hw) in [(2, 96, 160, 24), (8, 64, 256, 64), (8, 2048, 512, 128), (8, 512, 2048, 128), (8, 256, 1024, 128), (8, 512, 512, 256)]:
        g = torch.Generator().manual_seed(ci)
        x = torch.randn(n, ci, hw, hw, generator=g)
        x = (torch.relu(x) * (1 + x.abs())).cuda()
        dy = (torch.randn(n, co, hw, hw, generator=g) * 1e-5 * torch.exp(torch.randn(n, co, hw, hw, generator=g))).cuda()
        ref = torch.einsum('nohw,nchw->oc', dy[:, :32].double(), x[:, :32].double())
        outs = {}
        for name, fn in (('fp32', lambda d: H.conv_wgrad_(d, x, dy, 1)), ('bf16x6', lambda d: H.conv_wgrad_split_(d, x, dy, 1)),
                         ('f16x3', lambda d: H.conv_wgrad_f16x3_(d, x, dy, H.absmax(x), H.absmax(dy)))):
            d = torch.zeros(co, ci, 1, 1, device='cuda')
            fn(d)
            outs[name] = rel(d[:32, :32, 0, 0], ref)
        xa, da = H.absmax(x), H.absmax(dy)
        d = torch.zeros(co, ci, 1, 1, device='cuda')
        fl = 2.0 * n * ci * co * hw * hw
        t6 = timeit(lambda: H.conv_wgrad_split_(d, x, dy, 1))
        t3 = timeit(lambda: H.conv_wgrad_f16x3_(d, x, dy, xa, da))
        print(f'wgrad {n}x{ci}->{co}@{hw}: err fp32 {outs["fp32"]:.2e} bf16x6 {outs["bf16x6"]:.2e} f16x3 {outs["f16x3"]:.2e} | '
              f'bf16x6 {t6:.3f} ms ({fl / t6 / 1e9:4.0f} TF-eq)  f16x3 {t3:.3f} ms ({fl / t3 / 1e9:4.0f} TF-eq)', flush=True)


if __name__ == '__main__':
    if '--wgrad' in sys.argv:
        wgrad()
        sys.exit(0)
    if '--bench-only' not in sys.argv:
        accuracy()
    bench()
