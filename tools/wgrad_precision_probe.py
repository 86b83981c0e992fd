"""Where does the 1x1 weight gradient's error come from?  fp32-input MFMA kernel vs the bf16x6 split kernel against fp64, full dw,
for several operand distributions and pixel counts."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfst_amd import hip_ops as H      # noqa: E402


def rel(a, ref):
    return float((a.double() - ref).norm() / ref.norm())


def main():
    for (n, ci, co, hw) in [(8, 64, 64, 256), (8, 64, 64, 128), (8, 64, 64, 64), (2, 128, 128, 256), (8, 512, 128, 128)]:
        for dist in ('relu_tail x 1e-4 randn', 'randn x randn', 'uniform(1,2) x uniform(1,2)'):
            g = torch.Generator().manual_seed(1)
            x = torch.randn(n, ci, hw, hw, generator=g).cuda()
            dy = torch.randn(n, co, hw, hw, generator=g).cuda()
            if dist.startswith('relu'):
                x = torch.relu(x) * (1.0 + x.abs())
                dy = dy * 1e-4
            elif dist.startswith('uniform'):
                x = torch.rand(n, ci, hw, hw, generator=g).cuda() + 1.0
                dy = torch.rand(n, co, hw, hw, generator=g).cuda() + 1.0
            ref = torch.einsum('nohw,nchw->oc', dy.double(), x.double())
            d32 = torch.zeros(co, ci, 1, 1, device='cuda')
            H.conv_wgrad_(d32, x, dy, 1)
            d6 = torch.zeros(co, ci, 1, 1, device='cuda')
            H.conv_wgrad_split_(d6, x, dy, 1)
            # a plain fp32 torch reduction for scale: one long fp32 matmul
            dt = torch.einsum('nop,ncp->oc', dy.flatten(2), x.flatten(2))
            print(f'{n}x{ci}->{co}@{hw}^2  {dist:28s}  fp32-MFMA {rel(d32[:, :, 0, 0], ref):.2e}   bf16x6 {rel(d6[:, :, 0, 0], ref):.2e}   '
                  f'torch fp32 {rel(dt, ref):.2e}', flush=True)


if __name__ == '__main__':
    main()
