"""AdamW + poly LR schedule of the reference recipe (configs/_base_/schedules/adamw_40k.py:4-16) on the
flat parameter arena: one kernel launch per step for all 43.6 M parameters."""
import torch

from . import hip_ops as ops


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (amsgrad=False).  Parameters that live in a pfst_amd ParamArena are
    updated by ONE flat launch; any other CUDA parameter gets the same kernel per tensor."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = {}   # id(arena) -> dict(m, v, step)
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none=False):
        # gradients live in the arena; PFGST.forward_train zeroes it with one fill kernel
        for g in self.param_groups:
            for p in g['params']:
                if not hasattr(p, '_pfst_arena') and p.grad is not None:
                    p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            by_arena, loose = {}, []
            for p in group['params']:
                a = getattr(p, '_pfst_arena', None)
                (by_arena.setdefault(id(a), (a, []))[1] if a is not None else loose).append(p)
            for a, plist in by_arena.values():
                covered = sum((p.numel() + 3) // 4 * 4 for p in plist)
                if covered != a.numel:
                    loose.extend(plist)       # only part of the arena is optimised: per-tensor launches
                    continue
                if id(a) not in self._flat:
                    pend = getattr(self, '_pending_flat', None)
                    if pend:
                        p0 = pend.pop(0)
                        self._flat[id(a)] = dict(m=p0['m'].to(a.data.device), v=p0['v'].to(a.data.device), step=p0['step'])
                    else:
                        self._flat[id(a)] = dict(m=torch.zeros_like(a.data), v=torch.zeros_like(a.data), step=0)
                st = self._flat[id(a)]
                st['step'] += 1
                ops.adamw_step_(a.data, a.grad, st['m'], st['v'], group['lr'], group['betas'], group['eps'],
                                group['weight_decay'], st['step'], self.grad_scale)
            for p in loose:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st.update(m=torch.zeros_like(p.data), v=torch.zeros_like(p.data), step=0)
                st['step'] += 1
                ops.adamw_step_(p.data, p.grad.contiguous(), st['m'], st['v'], group['lr'], group['betas'], group['eps'],
                                group['weight_decay'], st['step'], self.grad_scale)


    def state_dict(self):
        sd = super().state_dict()
        sd['pfst_flat'] = [dict(m=st['m'].cpu(), v=st['v'].cpu(), step=st['step']) for st in self._flat.values()]
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        flat = sd.pop('pfst_flat', [])
        super().load_state_dict(sd)
        self._pending_flat = flat       # attached to the arenas on the first step() after the model is on the GPU


def build_optimizer(model, cfg):
    """rsiseg/core/builder.py:12-34 for the shipped config: AdamW over all requires_grad parameters."""
    cfg = dict(cfg)
    t = cfg.pop('type')
    if t != 'AdamW':
        raise NotImplementedError(f'optimizer {t}: the PFST recipe uses AdamW')
    cfg.pop('paramwise_cfg', None)
    return AdamW([p for p in model.parameters() if p.requires_grad], **cfg)


def poly_lr(base_lr, it, max_iters, power=1.0, min_lr=0.0, warmup_iters=1500, warmup_ratio=1e-6):
    """mmcv PolyLrUpdaterHook + linear warm-up (adamw_40k.py:8-16), by iteration."""
    coeff = (1 - it / max_iters) ** power
    lr = (base_lr - min_lr) * coeff + min_lr
    if it < warmup_iters:
        k = (1 - it / warmup_iters) * (1 - warmup_ratio)
        lr = lr * (1 - k)
    return lr
