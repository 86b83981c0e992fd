"""mIoU evaluation of the student: whole-image inference + confusion statistics on the GPU, the
metric arithmetic of rsiseg/core/evaluation/metrics.py:296-395 (`total_area_to_metrics`) on the host."""
from collections import OrderedDict

import numpy as np
import torch

from . import hip_ops as ops


class AreaAccumulator:
    """`intersect_and_union` summed over images (pre_eval=True path, metrics.py:26-86,296-330)."""

    def __init__(self, num_classes, ignore_index=255, device='cuda'):
        self.num_classes, self.ignore_index = num_classes, ignore_index
        self.hist = torch.zeros(3 * num_classes, dtype=torch.int64, device=device)

    def update(self, pred_u8, label):
        label8 = ops.to_u8(label.contiguous()) if label.dtype != torch.uint8 else label.contiguous()
        ops.confusion_hist_(self.hist, pred_u8.contiguous(), label8, self.num_classes, self.ignore_index)

    def areas(self):
        h = self.hist.cpu().numpy().astype(np.float64).reshape(3, self.num_classes)
        inter, pred, lab = h
        return inter, pred + lab - inter, pred, lab


def total_area_to_metrics(inter, union, pred, label, metrics=('mIoU',), nan_to_num=None, beta=1):
    out = OrderedDict(aAcc=inter.sum() / label.sum())
    with np.errstate(divide='ignore', invalid='ignore'):
        for m in metrics:
            if m == 'mIoU':
                out['IoU'] = inter / union
                out['Acc'] = inter / label
            elif m == 'mDice':
                out['Dice'] = 2 * inter / (pred + label)
                out['Acc'] = inter / label
            elif m == 'mFscore':
                p, r = inter / pred, inter / label
                out['Fscore'] = (1 + beta ** 2) * (p * r) / ((beta ** 2 * p) + r)
                out['Precision'], out['Recall'] = p, r
            else:
                raise KeyError(f'metrics {m} is not supported')
    if nan_to_num is not None:
        out = OrderedDict((k, np.nan_to_num(v, nan=nan_to_num)) for k, v in out.items())
    return out


@torch.no_grad()
def evaluate(model, batches, num_classes, ignore_index=255):
    """batches: iterable of dict(img=[N,C,H,W] cuda float, gt_semantic_seg=[N,1,H,W] or [N,H,W] integer labels).
    -> {'aAcc','mIoU','mAcc', 'IoU.<c>'...} in percent like CustomDataset.evaluate."""
    acc = None
    for b in batches:
        pred8, _ = model.inference(b['img'], b.get('img_metas'), rescale=False)
        if acc is None:
            acc = AreaAccumulator(num_classes, ignore_index, pred8.device)
        acc.update(pred8, b['gt_semantic_seg'])
    m = total_area_to_metrics(*acc.areas())
    res = OrderedDict(aAcc=100 * float(m['aAcc']), mIoU=100 * float(np.nanmean(m['IoU'])), mAcc=100 * float(np.nanmean(m['Acc'])))
    for c in range(num_classes):
        res[f'IoU.{c}'] = 100 * float(m['IoU'][c])
    return res


def revise_checkpoint_keys(state_dict, revise_keys=((r'^module\.', ''), ('model.', ''))):
    """mmcv.runner.load_checkpoint(revise_keys=...) as tools/test.py:237-242 calls it: every (pattern, replacement) is applied with
    re.sub to every key -- a PFGST checkpoint's `model.<student key>` becomes `<student key>`; the teacher's `ema_model.*`
    keys turn into names the segmentor does not have and are ignored (strict=False)."""
    import re
    from collections import OrderedDict
    out = OrderedDict()
    for k, v in state_dict.items():
        for pattern, repl in revise_keys:
            k = re.sub(pattern, repl, k)
        out[k] = v
    return out


def build_eval_fn(val_cfg, num_classes, device, metric='mIoU', max_images=None):
    """The validation pass the reference's EvalHook / DistEvalHook run every `evaluation.interval` iterations
    (rsiseg/apis/train.py:152-168, core/evaluation/eval_hooks.py:12-58, apis/test.py single_gpu_test with pre_eval): whole-tile
    inference of the student with samples_per_gpu = 1, per-class area statistics summed over the images (ranks take every
    world-th image and the statistics are all-reduced), then the metric table.  -> fn(model) -> {'aAcc', 'mIoU', 'mAcc', 'IoU.<c>'}"""
    import torch.distributed as dist
    from .data import TileFolder
    dataset = TileFolder(val_cfg, test_mode=True)
    metrics = (metric,) if isinstance(metric, str) else tuple(metric)

    @torch.no_grad()
    def run(model):
        seg = model.get_model() if hasattr(model, 'get_model') else model
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        if world > 1:
            # eval_hooks.py:95-107: rank 0's BatchNorm running statistics on every rank (of `runner.model`, i.e. student AND teacher),
            # so the summed histograms describe ONE model -- the one rank 0 checkpoints and tools/test.py scores
            from .dist import broadcast_bn_buffers_
            broadcast_bn_buffers_(model)
        acc = AreaAccumulator(num_classes, dataset.ignore_index, device)
        n = len(dataset) if max_images is None else min(len(dataset), max_images)
        for i in range(rank, n, world):
            item = dataset[i]
            pred8, _ = seg.inference(item['img'][None].to(device), [item['img_metas']], rescale=True)
            gt = torch.from_numpy(dataset.gt_seg_map(i)).to(device)
            acc.update(pred8.reshape(gt.shape), gt)
        if hasattr(seg, '_last_states'):
            seg._last_states = None          # do not keep the last tile's feature maps (~0.3 GB at 1024^2) alive through training
        if world > 1:
            dist.all_reduce(acc.hist)
        m = total_area_to_metrics(*acc.areas(), metrics=metrics)
        res = OrderedDict(aAcc=100 * float(m['aAcc']))
        for name, table in m.items():
            if name != 'aAcc':
                res['m' + name] = 100 * float(np.nanmean(table))
        for c in range(num_classes):
            res[f'IoU.{dataset.CLASSES[c] if c < len(dataset.CLASSES) else c}'] = 100 * float(m['IoU'][c]) if 'IoU' in m else float('nan')
        return res
    return run
