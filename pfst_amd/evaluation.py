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
