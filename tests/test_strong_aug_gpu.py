"""Strong-augmentation kernels (colour jitter, gaussian blur) against a PyTorch restatement of the SAME documented
kornia-0.6 formulas written here (kornia itself is not installed: parity with the reference's third-party arithmetic is
'unpinned'; this test pins the kernels to the restated formulas and checks their invariants)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
MEAN = torch.tensor([123.675, 116.28, 103.53]).view(1, 3, 1, 1)
STD = torch.tensor([58.395, 57.12, 57.375]).view(1, 3, 1, 1)


def rgb_to_hsv(img):
    mx, arg = img.max(1)
    mn = img.min(1)[0]
    d = mx - mn
    v = mx
    s = d / (mx + 1e-8)
    d = torch.where(d == 0, torch.ones_like(d), d)
    r, g, b = img[:, 0], img[:, 1], img[:, 2]
    rc, gc, bc = mx - r, mx - g, mx - b
    h = torch.stack([bc - gc, (rc - bc) + 2 * d, (gc - rc) + 4 * d], 1).gather(1, arg.unsqueeze(1)).squeeze(1)
    h = ((h / d) / 6.0) % 1.0
    return torch.stack([2 * math.pi * h, s, v], 1)


def hsv_to_rgb(hsv):
    h, s, v = hsv[:, 0] / (2 * math.pi), hsv[:, 1], hsv[:, 2]
    h6 = h * 6
    fl = torch.floor(h6)
    hi = fl.long() % 6
    f = h6 - fl
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    table = torch.stack([torch.stack([v, t, p], 1), torch.stack([q, v, p], 1), torch.stack([p, v, t], 1),
                         torch.stack([p, q, v], 1), torch.stack([t, p, v], 1), torch.stack([v, p, q], 1)], 1)   # [N,6,3,H,W]
    idx = hi.unsqueeze(1).unsqueeze(1).expand(-1, 1, 3, -1, -1)
    return table.gather(1, idx).squeeze(1)


def ref_jitter(img, prm):
    out = []
    for i in range(img.shape[0]):
        x = (img[i:i + 1] * STD + MEAN) / 255.0
        bf, cf, sf, hf = [float(v) for v in prm[i, :4]]
        for op in prm[i, 4:].long().tolist():
            if op == 0:
                x = (x + bf - 1).clamp(0, 1)
            elif op == 1:
                x = (x * cf).clamp(0, 1)
            else:
                hsv = rgb_to_hsv(x)
                if op == 2:
                    hsv[:, 1] = (hsv[:, 1] * sf).clamp(0, 1)
                else:
                    hsv[:, 0] = torch.remainder(hsv[:, 0] + hf, 2 * math.pi)
                x = hsv_to_rgb(hsv)
        out.append((x * 255.0 - MEAN) / STD)
    return torch.cat(out)


def test_color_jitter_matches_restated_formulas():
    from pfst_amd import hip_ops as ops
    g = torch.Generator().manual_seed(0)
    n, S = 3, 48
    img = torch.randn(n, 3, S, S, generator=g)
    prm = torch.tensor([[1.1, 0.9, 1.15, 0.3, 0, 1, 2, 3], [0.85, 1.2, 0.8, -0.7, 3, 2, 1, 0], [1.0, 1.0, 1.0, 0.0, 2, 0, 3, 1]])
    ref = ref_jitter(img, prm)
    out = ops.color_jitter_(img.clone().cuda(), prm.cuda(), MEAN.flatten().cuda(), STD.flatten().cuda(), True).cpu()
    err = (out - ref).abs()
    # hue arithmetic is discontinuous at sector borders: allow a handful of pixels to differ, the rest must be tight
    assert float((err > 1e-3).float().mean()) < 2e-3 and float(err.median()) < 1e-5
    # identity parameters leave in-gamut pixels unchanged
    ident = torch.tensor([[1.0, 1.0, 1.0, 0.0, 0, 1, 2, 3]])
    x = (torch.rand(1, 3, S, S, generator=g) * 255.0 - MEAN) / STD
    y = ops.color_jitter_(x.clone().cuda(), ident.cuda(), MEAN.flatten().cuda(), STD.flatten().cuda(), True).cpu()
    assert float((y - x).abs().max()) < 2e-4


def test_gaussian_blur_matches_conv_reference():
    from pfst_amd import hip_ops as ops
    from pfst_amd.strong_aug import _blur_kernel_size, _gauss_taps
    g = torch.Generator().manual_seed(1)
    n, H, W = 2, 96, 128
    img = torch.randn(n, 3, H, W, generator=g)
    ky, kx = _blur_kernel_size(H), _blur_kernel_size(W)
    assert ky % 2 == 1 and kx % 2 == 1 and (ky, kx) == (9, 13)
    sig = [0.4, 1.1]
    ty = torch.stack([_gauss_taps(ky, s) for s in sig])
    tx = torch.stack([_gauss_taps(kx, s) for s in sig])
    ref = []
    for i in range(n):
        x = F.pad(img[i:i + 1], (kx // 2, kx // 2, ky // 2, ky // 2), mode='reflect')
        x = F.conv2d(x, tx[i].view(1, 1, 1, kx).repeat(3, 1, 1, 1), groups=3)
        x = F.conv2d(x, ty[i].view(1, 1, ky, 1).repeat(3, 1, 1, 1), groups=3)
        ref.append(x)
    ref = torch.cat(ref)
    out = ops.gaussian_blur(img.cuda(), ty.cuda(), tx.cuda(), reach=13).cpu()
    assert float((out - ref).abs().max()) < 1e-5
    const = ops.gaussian_blur(torch.full((1, 3, 64, 64), 2.5).cuda(), ty[:1].cuda(), tx[:1].cuda(), reach=13).cpu()
    assert float((const - 2.5).abs().max()) < 1e-6          # taps sum to one


def test_strong_aug_entry_point_respects_draws():
    from pfst_amd.strong_aug import apply_strong_aug
    x = torch.randn(2, 3, 64, 64).cuda()
    metas = [dict(img_norm_cfg=dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375]))] * 2
    y = apply_strong_aug(x.clone(), metas, jitter_draw=0.1, jitter_p=0.2, jitter_s=0.2, blur_draw=0.3)
    assert torch.equal(y, x)                                  # neither draw exceeds its threshold -> untouched
    z = apply_strong_aug(x.clone(), metas, jitter_draw=0.9, jitter_p=0.2, jitter_s=0.2, blur_draw=0.9)
    assert z.shape == x.shape and bool(torch.isfinite(z).all()) and not torch.equal(z, x)
    ten = torch.randn(2, 10, 32, 32).cuda()
    assert apply_strong_aug(ten, metas, 0.9, 0.2, 0.2, 0.9) is ten   # != 3 channels: skipped like the reference
    import pytest
    with pytest.raises(ValueError, match='No such denorm type'):      # raised only when the jitter actually runs (dacs_transforms.py:69-74)
        apply_strong_aug(x.clone(), metas, 0.9, 0.2, 0.2, 0.3, denorm_type='minmax')
    assert torch.equal(apply_strong_aug(x.clone(), metas, 0.1, 0.2, 0.2, 0.3, denorm_type='minmax'), x)
    w01 = torch.rand(2, 3, 64, 64).cuda()                              # season_net: images already in [0, 1], no de-normalisation
    unit = [dict(img_norm_cfg=dict(mean=[0.0] * 3, std=[1.0] * 3, to_rgb=False))] * 2
    j = apply_strong_aug(w01.clone(), unit, 0.9, 0.2, 0.2, 0.3, denorm_type='none')
    assert bool(torch.isfinite(j).all()) and float(j.min()) >= -1e-6 and float(j.max()) <= 1.0 + 1e-6
