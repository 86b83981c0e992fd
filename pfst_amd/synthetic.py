"""Seeded synthetic inputs and weights shared by bench.py, the tests and the
golden-vector generator.  Pure torch-CPU helpers: no kernels, nothing from the oracle."""
import math

import torch

NORM_CFG = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375])


def synth_batch(b, S, num_classes, cin=3, seed=1234, block=None, device='cpu'):
    """One PFST batch dict (keys of rsiseg/datasets/uda_dataset.py:120-132 after collation):
    N(0,1) images, strong-aug = target + 0.1 N(0,1), blocky labels with a 255 ignore patch.

    Label blocks are 64 px ("repeat-interleave x64") from S = 128 up: PFGSTLoss's target-side mask needs all nine
    dilation-2 neighbours on the 1/8 grid un-mixed (pfgst_loss.py:64-71), which blocks of one or two grid pixels never give --
    with 8-px blocks `loss_sim_pos/neg` and their gradient were identically zero in every S=128 parity step."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(b, cin, S, S, generator=g)
    trg = torch.randn(b, cin, S, S, generator=g)
    trg_aug = trg + 0.1 * torch.randn(b, cin, S, S, generator=g)
    block = block or (min(64, S // 2) if S >= 128 else max(S // 16, 1))
    lab = torch.randint(0, num_classes, (b, 1, S // block, S // block), generator=g)
    gt = lab.repeat_interleave(block, 2).repeat_interleave(block, 3).contiguous()
    e = max(block // 8, 1) * 2
    gt[:, :, :e, :e] = 255
    mean = NORM_CFG['mean'] if cin == 3 else [120.0] * cin
    std = NORM_CFG['std'] if cin == 3 else [58.0] * cin
    metas = [dict(img_norm_cfg=dict(mean=list(mean), std=list(std))) for _ in range(b)]
    d = torch.device(device)
    return dict(img=img.to(d), img_metas=metas, gt_semantic_seg=gt.to(d), target_img=trg.to(d),
                target_img_metas=metas, target_img_strong_aug=trg_aug.to(d))


def fill_state_dict(sd, seed=0):
    """Overwrite every entry of a (reference-keyed) state_dict in place with seeded values that
    depend only on (seed, position, shape): kaiming-like conv weights, N(0, .01) classifier,
    BN gamma in [.6, 1.4], beta ~ .2 N, running stats reset.  Used so the reference run that made
    the golden vectors, the oracle and the HIP product all start from bit-identical weights."""
    i = -1
    for k, v in sd.items():
        if not torch.is_tensor(v):          # e.g. PFGST's `_extra_state`: not a reference key, no position
            continue
        i += 1
        g = torch.Generator().manual_seed(seed * 100003 + i)
        with torch.no_grad():
            if k.endswith('num_batches_tracked'):
                v.zero_()
            elif k.endswith('running_mean'):
                v.zero_()
            elif k.endswith('running_var'):
                v.fill_(1.0)
            elif v.dim() == 4:
                if 'conv_seg' in k:
                    v.copy_(torch.randn(v.shape, generator=g) * 0.01)
                else:
                    fan_in = v.shape[1] * v.shape[2] * v.shape[3]
                    v.copy_(torch.randn(v.shape, generator=g) * math.sqrt(2.0 / fan_in))
            elif 'conv_seg' in k:                       # classifier bias
                v.copy_(0.05 * torch.randn(v.shape, generator=g))
            elif k.endswith('.weight'):                 # BN gamma
                v.copy_(0.6 + 0.8 * torch.rand(v.shape, generator=g))
            else:                                       # BN beta
                v.copy_(0.2 * torch.randn(v.shape, generator=g))
    return sd
