"""Scratch probe (GPU box): pseudo-label near-tie behaviour of the kernel vs torch CPU / torch GPU softmax->max."""
import numpy as np
import torch
import torch.nn.functional as F

from pfst_amd import hip_ops as ops


def step_down(x, k):
    for _ in range(k):
        x = torch.nextafter(x, torch.full_like(x, -float('inf')))
    return x


def make(C, h, seed, mags):
    g = torch.Generator().manual_seed(seed)
    n = 2
    base = torch.randn(n, C, h, h, generator=g) * 0.5
    t = torch.tensor(mags)[torch.randint(0, len(mags), (n, 1, h, h), generator=g)]
    a = torch.randint(0, C, (n, 1, h, h), generator=g)
    b = (a + torch.randint(1, C, (n, 1, h, h), generator=g)) % C
    k = torch.randint(0, 4, (n, 1, h, h), generator=g)          # 0..3 ulp apart; 3 => all-equal pixel
    lo = t - 3.0 - base.abs()
    z = lo.clone()
    z.scatter_(1, a, t)
    zb = t.clone()
    for kk in (1, 2):
        zb = torch.where(k == kk, step_down(t, kk), zb)
    z.scatter_(1, b, zb)
    z = torch.where(k == 3, t.expand_as(z), z)
    return z.contiguous()


def ulps(a, b):
    return (a.numpy().view(np.int32).astype(np.int64) - b.numpy().view(np.int32).astype(np.int64))


for C in (2, 6, 33):
    for mags in ([0.01, 0.1, 0.5], [1.0, 3.0, 7.5], [-0.02, -1.0, -6.0], [0.003, 20.0, -40.0]):
        z = make(C, 64, 7 + C, mags)
        pc, lc = torch.max(torch.softmax(z, 1), 1)
        pg, lg = torch.max(torch.softmax(z.cuda(), 1), 1)
        l64, l8, cnt, prob = ops.pseudo_label(z.cuda(), (64, 64), 0.5, want_prob=True)
        am = z.argmax(1)
        print(f'C={C} mags={mags}: kernel!=cpu {(l64.cpu() != lc).sum().item()}  kernel!=gpu {(l64.cpu() != lg.cpu()).sum().item()}  '
              f'cpu!=gpu {(lc != lg.cpu()).sum().item()}  argmax(z)!=cpu {(am != lc).sum().item()} of {lc.numel()};  '
              f'prob ulp vs cpu {np.abs(ulps(prob.cpu(), pc)).max()}  vs gpu {np.abs(ulps(prob.cpu(), pg.cpu())).max()}')
        # upsampled x2
        up = F.interpolate(z, scale_factor=2, mode='bilinear', align_corners=False)
        pc, lc = torch.max(torch.softmax(up, 1), 1)
        upg = F.interpolate(z.cuda(), scale_factor=2, mode='bilinear', align_corners=False)
        pg, lg = torch.max(torch.softmax(upg, 1), 1)
        l64, l8, cnt, prob = ops.pseudo_label(z.cuda(), (128, 128), 0.5, want_prob=True)
        print(f'    x2: kernel!=cpu {(l64.cpu() != lc).sum().item()}  kernel!=gpu {(l64.cpu() != lg.cpu()).sum().item()}  cpu!=gpu {(lc != lg.cpu()).sum().item()}'
              f'  up cpu!=gpu elems {(up != upg.cpu()).sum().item()} of {up.numel()};  prob ulp vs cpu {np.abs(ulps(prob.cpu(), pc)).max()} vs gpu {np.abs(ulps(prob.cpu(), pg.cpu())).max()}')
# random logits: probability ulp distance and threshold count
z = torch.randn(2, 6, 64, 64, generator=torch.Generator().manual_seed(1)) * 4
up = F.interpolate(z, size=(256, 256), mode='bilinear', align_corners=False)
pc, lc = torch.max(torch.softmax(up, 1), 1)
pg, lg = torch.max(torch.softmax(F.interpolate(z.cuda(), size=(256, 256), mode='bilinear', align_corners=False), 1), 1)
l64, l8, cnt, prob = ops.pseudo_label(z.cuda(), (256, 256), 0.9, want_prob=True)
d = ulps(prob.cpu(), pc)
print('random logits: labels equal cpu', torch.equal(l64.cpu(), lc), 'gpu', torch.equal(l64.cpu(), lg.cpu()), ' prob ulp vs cpu hist', np.unique(d, return_counts=True),
      ' vs gpu max', np.abs(ulps(prob.cpu(), pg.cpu())).max(), ' count', int(cnt), int((pc >= 0.9).sum()), int((pg >= 0.9).sum()))
