"""Time pfst_bn_backward on the train step's plane shapes; prints GB/s against the algorithmic bytes (read dy, x [, y],
write dx [, dres]).  Use PFST_HIP_LIB=<other libpfst_hip.so> for same-box A/B runs of kernel variants."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfst_amd import hip_ops as H      # noqa: E402

SHAPES = [(2, 64, 512, 512, False), (2, 64, 256, 256, False), (2, 256, 256, 256, True), (2, 128, 128, 128, False),
          (2, 512, 128, 128, True), (2, 256, 128, 128, False), (2, 1024, 128, 128, True), (2, 512, 128, 128, False),
          (2, 2048, 128, 128, True), (2, 256, 256, 256, False), (2, 48, 256, 256, False), (2, 304, 256, 256, False)]


def main():
    dev = 'cuda'
    tot = 0.0
    for n, c, h, w, res in SHAPES:
        x = torch.randn(n, c, h, w, device=dev)
        dy = torch.randn_like(x)
        g = torch.rand(c, device=dev) + 0.5
        b = torch.randn(c, device=dev) * 0.1
        mean, invstd = H.bn_stats(x)
        r = torch.randn_like(x) if res else None
        y = H.bn_apply(x, mean, invstd, g, b, relu=True, residual=r)
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        dres = torch.zeros_like(x) if res else None
        dx = torch.empty_like(x)

        def run():
            H.bn_backward(dy, y if res else None, x, mean, invstd, g, dg, db, relu=True, dres=dres, dres_accumulate=res, dx=dx,
                          beta=b)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        nbytes = 4.0 * x.numel() * (3 + (3 if res else 0))     # residual form: + y read, dres read + write
        tot += ms
        print(f'{n}x{c}x{h}x{w} res={int(res)}  {ms:7.3f} ms  {nbytes / ms / 1e6:7.0f} GB/s', flush=True)
    print(f'total {tot:.3f} ms')


if __name__ == '__main__':
    main()
