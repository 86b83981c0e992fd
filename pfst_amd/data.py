"""Batch sources for the train step.  The reference's CPU data pipeline (rsiseg/datasets) is out
of scope for the kernels; what the hot path needs is the batch CONTRACT of UDADataset after collation
(rsiseg/datasets/uda_dataset.py:116-135): img / img_metas / gt_semantic_seg / target_img / target_img_metas /
target_img_strong_aug.  Two sources: seeded synthetic tensors (benchmarks, tests) and a minimal ISPRS tile reader."""
import os
import random

import numpy as np
import torch

from .synthetic import NORM_CFG, synth_batch


def synthetic_loader(batch_size, size, num_classes, in_channels=3, seed=1234, device='cuda', distinct=4):
    """infinite iterator cycling over `distinct` pre-generated device batches"""
    batches = [synth_batch(batch_size, size, num_classes, in_channels, seed=seed + i, device=device) for i in range(distinct)]
    i = 0
    while True:
        yield batches[i % distinct]
        i += 1


class ISPRSTiles:
    """1024x1024 PNG tiles produced by tools/convert_datasets/potsdam.py / vaihingen.py (img_dir/*.png, ann_dir/*.png)."""

    def __init__(self, img_dir, ann_dir=None, crop_size=(512, 512), ratio_range=(0.5, 2.0), cat_max_ratio=0.75,
                 img_scale=(1024, 1024), mean=NORM_CFG['mean'], std=NORM_CFG['std'], train=True):
        from PIL import Image  # noqa: F401
        self.img_dir, self.ann_dir = img_dir, ann_dir
        self.files = sorted(f for f in os.listdir(img_dir) if f.lower().endswith(('.png', '.jpg', '.tif')))
        self.crop_size, self.ratio_range, self.cat_max_ratio, self.img_scale = crop_size, ratio_range, cat_max_ratio, img_scale
        self.mean, self.std = np.array(mean, np.float32), np.array(std, np.float32)
        self.train = train

    def __len__(self):
        return len(self.files)

    def _load(self, idx):
        from PIL import Image
        name = self.files[idx]
        img = Image.open(os.path.join(self.img_dir, name)).convert('RGB')
        ann = Image.open(os.path.join(self.ann_dir, os.path.splitext(name)[0] + '.png')) if self.ann_dir else None
        return img, ann

    def get(self, idx, strong=False):
        """Resize(ratio 0.5-2) -> RandomCrop(cat_max_ratio) -> RandomRotate90 -> flips -> Normalize
        (configs/_base_/datasets/pots_irrg2vaih_irrg.py:11-45); `strong` adds a brightness/contrast perturbation."""
        from PIL import Image
        img, ann = self._load(idx)
        if self.train:
            r = random.uniform(*self.ratio_range)
            size = (int(self.img_scale[0] * r + 0.5), int(self.img_scale[1] * r + 0.5))
            img = img.resize(size, Image.BILINEAR)
            ann = ann.resize(size, Image.NEAREST) if ann is not None else None
        im = np.asarray(img, np.float32)
        lab = np.asarray(ann, np.uint8) if ann is not None else np.full(im.shape[:2], 255, np.uint8)
        if self.train:
            ch, cw = self.crop_size
            ph, pw = max(ch - im.shape[0], 0), max(cw - im.shape[1], 0)
            if ph or pw:
                im = np.pad(im, ((0, ph), (0, pw), (0, 0)))
                lab = np.pad(lab, ((0, ph), (0, pw)), constant_values=255)
            for _ in range(10):
                y = random.randint(0, im.shape[0] - ch)
                x = random.randint(0, im.shape[1] - cw)
                crop = lab[y:y + ch, x:x + cw]
                vals, cnt = np.unique(crop[crop != 255], return_counts=True)
                if len(cnt) > 1 and cnt.max() / cnt.sum() < self.cat_max_ratio:
                    break
            im, lab = im[y:y + ch, x:x + cw], lab[y:y + ch, x:x + cw]
            k = random.randint(0, 3)
            im, lab = np.rot90(im, k), np.rot90(lab, k)
            if random.random() < 0.5:
                im, lab = im[:, ::-1], lab[:, ::-1]
            if random.random() < 0.5:
                im, lab = im[::-1], lab[::-1]
        aug = None
        if strong:
            aug = np.clip(im * random.uniform(0.6, 1.4) + random.uniform(-32, 32), 0, 255)
            aug = (aug - self.mean) / self.std
        im = (im - self.mean) / self.std
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return t(im), torch.from_numpy(np.ascontiguousarray(lab))[None], (t(aug) if aug is not None else None)


def get_rcs_class_probs(data_root, temperature):
    """Rare-class-sampling distribution (rsiseg/datasets/uda_dataset.py:17-40): classes sorted by total pixel count,
    p = softmax((1 - freq) / T) in float32 as torch computes it."""
    import json
    with open(os.path.join(data_root, 'sample_class_stats.json')) as f:
        sample_class_stats = json.load(f)
    overall = {}
    for s in sample_class_stats:
        s.pop('file')
        for c, n in s.items():
            overall[int(c)] = overall.get(int(c), 0) + n
    overall = dict(sorted(overall.items(), key=lambda item: item[1]))
    freq = torch.tensor(list(overall.values()))
    freq = freq / torch.sum(freq)
    freq = torch.softmax((1 - freq) / temperature, dim=-1)
    return list(overall.keys()), freq.numpy()


class UDADataset:
    """Source/target pairing of the reference (uda_dataset.py:43-135): sample idx = (source[idx // len(target)],
    target[idx % len(target)]); with `rare_class_sampling` a class is drawn from get_rcs_class_probs, then a source file
    containing it, re-cropped up to 10 times until it holds enough pixels of that class, then a uniform target -- the
    same NumPy RNG calls in the same order, so a seeded run picks the same samples as the reference."""

    def __init__(self, source, target, cfg):
        import json
        self.source, self.target = source, target
        self.ignore_index = target.ignore_index
        self.CLASSES, self.PALETTE = target.CLASSES, target.PALETTE
        self.path2name = cfg.get('path2name', False)
        assert target.ignore_index == source.ignore_index and target.CLASSES == source.CLASSES and target.PALETTE == source.PALETTE
        rcs = cfg.get('rare_class_sampling')
        self.rcs_enabled = rcs is not None
        if self.rcs_enabled:
            self.rcs_class_temp, self.rcs_min_crop_ratio, self.rcs_min_pixels = rcs['class_temp'], rcs['min_crop_ratio'], rcs['min_pixels']
            root = cfg['source'].get('rcs_root', cfg['source']['data_root'])
            self.rcs_classes, self.rcs_classprob = get_rcs_class_probs(root, self.rcs_class_temp)
            with open(os.path.join(root, 'samples_with_class.json')) as f:
                swc = {int(k): v for k, v in json.load(f).items() if int(k) in self.rcs_classes}
            self.samples_with_class = {}
            for c in self.rcs_classes:
                self.samples_with_class[c] = [file.split('/')[-1] for file, pixels in swc[c] if pixels > self.rcs_min_pixels]
                assert len(self.samples_with_class[c]) > 0
            self.file_to_idx = {}
            for i, info in enumerate(self.source.img_infos):
                file = info['ann']['seg_map']
                self.file_to_idx[file.split('/')[-1] if self.path2name else file] = i

    def get_rare_class_sample(self):
        c = np.random.choice(self.rcs_classes, p=self.rcs_classprob)
        f1 = np.random.choice(self.samples_with_class[c])
        i1 = self.file_to_idx[f1]
        s1 = self.source[i1]
        if self.rcs_min_crop_ratio > 0:
            for _ in range(10):
                if torch.sum(s1['gt_semantic_seg'].data == c) > self.rcs_min_pixels * self.rcs_min_crop_ratio:
                    break
                s1 = self.source[i1]                  # a new random crop of the same source image
        s2 = self.target[np.random.choice(range(len(self.target)))]
        return {**s1, 'target_img_metas': s2['img_metas'], 'target_img': s2['img']}

    def __getitem__(self, idx):
        if self.rcs_enabled:
            return self.get_rare_class_sample()
        s1 = self.source[idx // len(self.target)]
        s2 = self.target[idx % len(self.target)]
        out = {**s1, 'target_img_metas': s2['img_metas'], 'target_img': s2['img']}
        if 'img_strong_aug' in s2:
            out['target_img_strong_aug'] = s2['img_strong_aug']
        if 'ori_img' in s2:
            out['target_img_ori'] = s2['ori_img']
        return out

    def __len__(self):
        return len(self.source) * len(self.target)


def uda_loader(source, target, batch_size, device='cuda', seed=0, rank=0, world=1):
    """UDADataset pairing (uda_dataset.py:116-135: idx // len(target), idx % len(target)) -> collated device batches."""
    rng = random.Random(seed + rank)
    meta = dict(img_norm_cfg=dict(mean=list(NORM_CFG['mean']), std=list(NORM_CFG['std'])))
    n = len(source) * len(target)
    while True:
        imgs, gts, timgs, taugs = [], [], [], []
        for _ in range(batch_size):
            idx = rng.randrange(n)
            s_img, s_gt, _ = source.get(idx // len(target))
            t_img, _, t_aug = target.get(idx % len(target), strong=True)
            imgs.append(s_img), gts.append(s_gt), timgs.append(t_img), taugs.append(t_aug)
        metas = [dict(meta) for _ in range(batch_size)]
        yield dict(img=torch.stack(imgs).to(device), img_metas=metas, gt_semantic_seg=torch.stack(gts).to(device),
                   target_img=torch.stack(timgs).to(device), target_img_metas=metas,
                   target_img_strong_aug=torch.stack(taugs).to(device))
