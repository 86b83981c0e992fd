"""Iteration-based trainer glue: the slice of mmcv's IterBasedRunner + hooks that
rsiseg/apis/train.py:71-192 configures for PFST -- poly LR with linear warm-up, text logging of the averaged
`log_vars` every `log_config.interval`, checkpoints every `checkpoint_config.interval` (+ resume incl. `local_iter`,
which the reference forgets), optional mIoU evaluation every `evaluation.interval`."""
import json
import os
import random
import time
from collections import OrderedDict, defaultdict

import numpy as np
import torch
import torch.distributed as dist

from .optim import poly_lr


def set_random_seed(seed, deterministic=False):
    """rsiseg/apis/train.py:52-68.  `deterministic` (the reference: cudnn.deterministic = True, benchmark = False, i.e. run-to-run reproducible
    kernels) switches the kernel library to its fixed-order mode (hip_ops.set_deterministic): the split-K slices of the weight gradients and the
    BatchNorm-backward / depthwise / bias reductions go through per-workgroup partials summed in index order -- no sum depends on the order in which
    workgroups finish, the gradient of a step is bit-identical between runs and stream schedules.  About 3 % slower; without the flag runs are
    reproducible up to that summation order only (two runs of one full-size step differ by up to 5e-2 norm-wise on the gradient after ~70 train-mode BatchNorm layers)."""
    if deterministic:
        from . import hip_ops
        hip_ops.set_deterministic(True)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def init_random_seed(seed=None, device='cuda'):
    """rsiseg/apis/train.py:21-49: rank 0 draws, everybody receives."""
    if seed is not None:
        return seed
    seed = np.random.randint(2 ** 31)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor(seed if dist.get_rank() == 0 else 0, dtype=torch.int32, device=device)
        dist.broadcast(t, src=0)
        seed = int(t.item())
    return seed


def find_latest_checkpoint(path, suffix='pth'):
    """rsiseg/utils/misc.py:7-41"""
    if not os.path.isdir(path):
        return None
    latest = os.path.join(path, f'latest.{suffix}')
    if os.path.exists(latest):
        return latest
    best, best_it = None, -1
    for f in os.listdir(path):
        if f.startswith('iter_') and f.endswith('.' + suffix):
            it = int(f[5:-len(suffix) - 1])
            if it > best_it:
                best, best_it = os.path.join(path, f), it
    return best


def is_main():
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


class IterBasedRunner:
    def __init__(self, model, optimizer, cfg, work_dir=None, eval_fn=None, log=print):
        self.model, self.optimizer, self.cfg, self.work_dir, self.eval_fn, self.log = model, optimizer, cfg, work_dir, eval_fn, log
        self.max_iters = cfg.runner.max_iters
        self.iter = 0
        self.epoch = 0
        self.base_lr = [g['lr'] for g in optimizer.param_groups]
        lr_cfg = cfg.get('lr_config') or {}
        if lr_cfg.get('policy', 'poly') != 'poly' or lr_cfg.get('warmup', 'linear') not in (None, 'linear'):
            raise NotImplementedError(f'lr_config {lr_cfg}: the PFST recipe is poly + linear warm-up')
        self.lr_cfg = dict(power=lr_cfg.get('power', 1.0), min_lr=lr_cfg.get('min_lr', 0.0),
                           warmup_iters=lr_cfg.get('warmup_iters', 0) if lr_cfg.get('warmup') else 0,
                           warmup_ratio=lr_cfg.get('warmup_ratio', 1e-6))
        self.log_interval = (cfg.get('log_config') or {}).get('interval', 50)
        self.ckpt_interval = (cfg.get('checkpoint_config') or {}).get('interval', 0)
        self.eval_interval = (cfg.get('evaluation') or {}).get('interval', 0)
        self.buffer = defaultdict(list)
        if work_dir and is_main():
            os.makedirs(work_dir, exist_ok=True)

    # ---- checkpointing (mmcv CheckpointHook layout: meta / state_dict / optimizer)
    def save_checkpoint(self):
        if not (self.work_dir and is_main()):
            return None
        path = os.path.join(self.work_dir, f'iter_{self.iter}.pth')
        # meta as mmcv's runner writes / reads it back on resume (epoch + iter) plus what tools/train.py:228-236 adds (CLASSES, PALETTE)
        meta = dict(iter=self.iter, epoch=self.epoch, time=time.asctime(), CLASSES=getattr(self.model, 'CLASSES', None),
                    PALETTE=getattr(self.model, 'PALETTE', None))
        torch.save(dict(meta=meta,
                        state_dict=OrderedDict((k, v.cpu() if torch.is_tensor(v) else v) for k, v in self.model.state_dict().items()),
                        optimizer=self.optimizer.state_dict()), path)
        latest = os.path.join(self.work_dir, 'latest.pth')
        if os.path.lexists(latest):
            os.remove(latest)
        os.symlink(os.path.basename(path), latest)
        return path

    def resume(self, path):
        ckpt = torch.load(path, map_location='cpu', weights_only=False)
        self.model.load_state_dict(ckpt['state_dict'], strict=False)
        if 'optimizer' in ckpt:
            self.optimizer.load_state_dict(ckpt['optimizer'])
        self.iter = ckpt.get('meta', {}).get('iter', 0)
        self.epoch = ckpt.get('meta', {}).get('epoch', 0)         # build_loader(start_epoch=runner.epoch) continues the data order
        self.log(f'resumed from {path} at iter {self.iter}')

    def load_checkpoint(self, path, revise_keys=(('module.', ''),)):
        ckpt = torch.load(path, map_location='cpu', weights_only=False)
        sd = ckpt.get('state_dict', ckpt)
        for a, b in revise_keys:
            sd = OrderedDict((k[len(a):] if k.startswith(a) else k, v) for k, v in sd.items())
        return self.model.load_state_dict(sd, strict=False)

    # ---- the loop
    def current_lr(self):
        return [poly_lr(b, self.iter, self.max_iters, **self.lr_cfg) for b in self.base_lr]

    def run(self, data_iter, max_iters=None):
        max_iters = max_iters or self.max_iters
        t_last = time.time()
        while self.iter < max_iters:
            t0 = time.time()
            batch = next(data_iter)
            data_time = time.time() - t0
            self.epoch = getattr(data_iter, 'epoch', self.epoch)          # meta['epoch'] of the checkpoints: the epoch of the CONSUMED data
            for g, lr in zip(self.optimizer.param_groups, self.current_lr()):
                g['lr'] = lr
            out = self.model.train_step(batch, self.optimizer)
            self.iter += 1
            for k, v in out['log_vars'].items():
                self.buffer[k].append(v)
            self.buffer['data_time'].append(data_time)
            if self.iter % self.log_interval == 0 and is_main():
                now = time.time()
                it_time = (now - t_last) / self.log_interval
                t_last = now
                avg = OrderedDict((k, float(np.mean(v))) for k, v in self.buffer.items())
                eta = it_time * (max_iters - self.iter)
                mem = torch.cuda.max_memory_allocated() // (1024 * 1024) if torch.cuda.is_available() else 0
                data_time = avg.pop('data_time')
                msg = (f'Iter [{self.iter}/{max_iters}]\tlr: {self.optimizer.param_groups[0]["lr"]:.3e}, eta: {eta / 3600:.2f} h, '
                       f'time: {it_time:.3f}, data_time: {data_time:.3f}, memory: {mem}, ')
                msg += ', '.join(f'{k}: {v:.4f}' for k, v in avg.items())
                self.log(msg)
                if self.work_dir:
                    with open(os.path.join(self.work_dir, 'log.json'), 'a') as f:
                        f.write(json.dumps(dict(mode='train', iter=self.iter, lr=self.optimizer.param_groups[0]['lr'],
                                                time=it_time, data_time=data_time, memory=mem, **avg)) + '\n')     # mmcv's TextLoggerHook keys
                self.buffer.clear()
            if self.ckpt_interval and self.iter % self.ckpt_interval == 0:
                self.save_checkpoint()
            if self.eval_fn and self.eval_interval and self.iter % self.eval_interval == 0:
                res = self.eval_fn(self.model)
                if is_main():
                    self.log(f'Iter(val) [{self.iter}]\t' + ', '.join(f'{k}: {v:.2f}' for k, v in res.items() if not k.startswith('IoU.')))
                    if self.work_dir:
                        with open(os.path.join(self.work_dir, 'log.json'), 'a') as f:
                            f.write(json.dumps(dict(mode='val', iter=self.iter, **res)) + '\n')
        return self.iter
