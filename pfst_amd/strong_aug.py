"""Strong augmentation of the class-mixed image (rsiseg/models/utils/dacs_transforms.py:12-27,44-107): colour jitter
(if the step's U(0,1) draw > p) then gaussian blur (if the step's draw > 0.5), only for 3-channel images.
The per-image random parameters are drawn on the host exactly where the reference draws them (NumPy for the blur
sigma; torch's CPU generator stands in for kornia's sampler); the pixels are processed by HIP kernels.
kornia's arithmetic is restated from its documentation -- parity unpinned."""
import math

import numpy as np
import torch

from . import hip_ops as ops

AVAILABLE = True


def _blur_kernel_size(n):
    return int(np.floor(np.ceil(0.1 * n) - 0.5 + np.ceil(0.1 * n) % 2))


def _gauss_taps(k, sigma):
    x = torch.arange(k, dtype=torch.float32) - k // 2
    if k % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2) / (2.0 * sigma * sigma))
    return g / g.sum()


def apply_strong_aug(mixed_img, img_metas, jitter_draw, jitter_p, jitter_s, blur_draw, denorm_type='mean_std'):
    n, c, h, w = mixed_img.shape
    if c != 3:
        return mixed_img
    dev = mixed_img.device
    if jitter_draw > jitter_p:
        if denorm_type not in ('mean_std', 'none'):          # dacs_transforms.py:69-74, raised where the reference raises it
            raise ValueError('No such denorm type!')
        s = jitter_s if isinstance(jitter_s, dict) else dict(brightness=jitter_s, contrast=jitter_s, saturation=jitter_s, hue=jitter_s)
        prm = torch.empty(n, 8)
        for i in range(n):                      # the reference builds one ColorJitter per image (pfgst.py:287-294)
            b = torch.empty(1).uniform_(max(0.0, 1 - s['brightness']), 1 + s['brightness'])
            ct = torch.empty(1).uniform_(max(0.0, 1 - s['contrast']), 1 + s['contrast'])
            sa = torch.empty(1).uniform_(max(0.0, 1 - s['saturation']), 1 + s['saturation'])
            hu = torch.empty(1).uniform_(-s['hue'], s['hue']) * 2 * math.pi
            prm[i] = torch.cat([b, ct, sa, hu, torch.randperm(4).float()])
        meta = img_metas[0]['img_norm_cfg']
        mean = ops.const_tensor(meta['mean'], dev)
        std = ops.const_tensor(meta['std'], dev)
        ops.color_jitter_(mixed_img, ops.h2d_small(prm, dev, 'jitter'), mean, std, denorm_type == 'mean_std')
    if blur_draw > 0.5:
        ky, kx = _blur_kernel_size(h), _blur_kernel_size(w)
        ty, tx = torch.empty(n, ky), torch.empty(n, kx)
        smax = 0.0
        for i in range(n):
            sigma = np.random.uniform(0.15, 1.15)    # one NumPy draw per image, as the reference
            smax = max(smax, sigma)
            ty[i], tx[i] = _gauss_taps(ky, sigma), _gauss_taps(kx, sigma)
        reach = int(math.ceil(smax * 10.0)) + 1      # exp(-50) ~ 2e-22: dropped taps are below fp32 resolution
        mixed_img = ops.gaussian_blur(mixed_img, ops.h2d_small(ty, dev, 'blur_y'), ops.h2d_small(tx, dev, 'blur_x'), reach)
    return mixed_img
