"""The worker-process loader and the native pixel kernels of the data path.

  * csrc/pipeline_cpu.c (libpfst_cpu.so) gives the same pixels bit for bit as the NumPy restatement in pfst_amd/pipeline.py, and the
    lazily evaluated Resize -> RandomCrop window equals the eager resize followed by the crop;
  * build_loader: identical batches for 0 and 2 worker processes given the seed (one RNG stream per sample), the training thread's
    global NumPy / Python streams untouched by inline loading, the reference's worker_init_fn seeding as an option;
  * rank sharding: two ranks draw disjoint index sets of the same permutation."""
import os
import random

import numpy as np
import pytest
import torch

from test_data_pipeline_cpu import NORM, SOURCE, TARGET, _tile


def _folders(root, n_src=4, n_trg=3, size=256):
    from PIL import Image
    for dom, n in (('pots', n_src), ('vaih', n_trg)):
        os.makedirs(root / dom / 'img_dir/train'), os.makedirs(root / dom / 'ann_dir/train')
        for i in range(n):
            img, seg = _tile(10 * n + i, size)
            Image.fromarray(img).save(root / dom / 'img_dir/train' / f't{i}.png')
            Image.fromarray(seg).save(root / dom / 'ann_dir/train' / f't{i}.png')
    loader = dict(reduce_zero_label=True)
    return dict(type='UDADataset',
                source=dict(type='ISPRSDataset', data_root=str(root / 'pots'), img_dir='img_dir/train', ann_dir='ann_dir/train',
                            gt_seg_map_loader_cfg=loader, pipeline=SOURCE),
                target=dict(type='ISPRSDataset', data_root=str(root / 'vaih'), img_dir='img_dir/train', ann_dir='ann_dir/train',
                            gt_seg_map_loader_cfg=loader, pipeline=TARGET), rare_class_sampling=None)


def test_native_pixel_kernels_equal_the_numpy_restatement():
    from pfst_amd import pipeline as PL
    from pfst_amd.build import build_cpu
    build_cpu(verbose=False)
    assert PL.set_native(True), 'libpfst_cpu.so was not built / loaded'
    rng = np.random.RandomState(0)
    px = rng.randint(0, 256, (200000, 1, 3)).astype(np.uint8)
    px[:256, 0, :] = np.arange(256)[:, None]                       # greys (d == 0), black, white
    px[256:512, 0, 0] = np.arange(256)                             # every hue byte, also the ones >= 180 a uint8 can hold
    assert np.array_equal(PL.bgr2hsv_u8(px), PL.bgr2hsv_np(px))
    assert np.array_equal(PL.hsv2bgr_u8(px), PL.hsv2bgr_np(px))
    img, seg = _tile(3, 384)
    outs = {}
    try:
        for nat in (False, True):
            PL.set_native(nat)
            for lazy in (False, True):
                PL.Pipeline.lazy_resize = lazy
                np.random.seed(11)
                res = []
                for _ in range(6):                                 # several draws: ratios above and below 1, every photometric branch
                    res.append(PL.Pipeline(SOURCE)(img, seg))
                    res.append(PL.Pipeline(TARGET)(img, None))
                outs[(nat, lazy)] = (res, np.random.get_state()[1][:8].copy())
    finally:
        PL.Pipeline.lazy_resize = True
        PL.set_native(True)
    base, base_state = outs[(False, False)]                        # NumPy, eager resize: the restatement
    for key, (res, state) in outs.items():
        assert np.array_equal(state, base_state), key              # the same draws from the NumPy stream
        for a, b in zip(base, res):
            assert set(a) == set(b)
            for k in a:
                if isinstance(a[k], np.ndarray):
                    assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), (key, k)


def _first_batches(loader, n):
    out = []
    for _ in range(n):
        b = next(loader)
        out.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()})
    loader.close()
    return out


def test_batches_do_not_depend_on_the_worker_count(tmp_path):
    from pfst_amd.data import build_loader, build_uda_dataset
    ds = build_uda_dataset(_folders(tmp_path))
    np.random.seed(123); random.seed(123)
    np_before, py_before = np.random.get_state()[1].copy(), random.getstate()
    inline = _first_batches(build_loader(ds, 2, device='cpu', seed=7, workers=0), 4)
    # inline loading leaves the training thread's global streams alone (PFGST draws its class mix from the NumPy one)
    assert np.array_equal(np.random.get_state()[1], np_before) and random.getstate() == py_before
    workers = _first_batches(build_loader(ds, 2, device='cpu', seed=7, workers=2), 4)
    for a, b in zip(inline, workers):
        assert set(a) == {'img', 'gt_semantic_seg', 'target_img', 'target_img_strong_aug', 'img_metas', 'target_img_metas'}
        for k in ('img', 'gt_semantic_seg', 'target_img', 'target_img_strong_aug'):
            assert torch.equal(a[k], b[k]), k
        assert [m['filename'] for m in a['img_metas']] == [m['filename'] for m in b['img_metas']]
    assert inline[0]['img'].shape == (2, 3, 512, 512) and inline[0]['gt_semantic_seg'].dtype == torch.uint8
    other_seed = _first_batches(build_loader(ds, 2, device='cpu', seed=8, workers=0), 1)
    assert not torch.equal(other_seed[0]['img'], inline[0]['img'])
    # the per-sample streams differ between samples and between epochs
    from pfst_amd.data import sample_seed
    seeds = {sample_seed(7, e, p, r, 2) for e in range(3) for p in range(50) for r in range(2)}
    assert len(seeds) == 300


def test_reference_worker_seeding_option_and_rank_sharding(tmp_path):
    from pfst_amd.data import _worker_init, build_loader, build_uda_dataset, epoch_indices
    ds = build_uda_dataset(_folders(tmp_path))
    # seeding='worker' inline = worker 0 of 0: the stream is seeded num_workers * rank + worker_id + seed (builder.py:170-181)
    saved = np.random.get_state()
    try:
        a = _first_batches(build_loader(ds, 2, device='cpu', seed=5, rank=1, world=2, workers=0, seeding='worker'), 1)
        np.random.seed(0 * 1 + 0 + 5); random.seed(5)
        idx = epoch_indices(len(ds), 2, 1, 0, 5)[:2]
        want = [ds[i] for i in idx]
        assert torch.equal(a[0]['img'], torch.stack([w['img'] for w in want]))
    finally:
        np.random.set_state(saved)
    r0 = epoch_indices(len(ds), 2, 0, 0, 5)
    r1 = epoch_indices(len(ds), 2, 1, 0, 5)
    assert set(r0).isdisjoint(r1) and sorted(r0 + r1) == list(range(len(ds)))
    with pytest.raises(ValueError):
        build_loader(ds, 2, device='cpu', seeding='nope')
