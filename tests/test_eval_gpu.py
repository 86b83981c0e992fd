"""Test-time path and trainer glue (f1) on the GPU: whole-image inference in eval mode vs the oracle,
confusion statistics / mIoU vs NumPy, checkpoint + resume of a short synthetic run."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import seeded_pfgst_state, uda_cfg

pytestmark = pytest.mark.gpu


def test_inference_eval_mode_and_miou():
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd.evaluation import AreaAccumulator, evaluate, total_area_to_metrics
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import synth_batch
    model = UDA.build(uda_cfg())
    both, student, _ = seeded_pfgst_state(O, 9)
    g = torch.Generator().manual_seed(1)
    for k, v in both.items():                      # non-trivial running statistics
        if k.endswith('running_mean'):
            v.copy_(0.05 * torch.randn(v.shape, generator=g))
        elif k.endswith('running_var'):
            v.copy_(0.8 + 0.4 * torch.rand(v.shape, generator=g))
    student = {k[6:]: v for k, v in both.items() if k.startswith('model.')}
    model.load_state_dict(both, strict=False)
    model.cuda()
    batch = synth_batch(2, 128, 6, seed=5)
    with torch.no_grad():
        feats = O.backbone_forward(student, batch['img'], train=False)
        logits, _ = O.decode_head_forward(student, feats, train=False)
        ref = F.interpolate(logits, size=(128, 128), mode='bilinear', align_corners=False).softmax(1).argmax(1)
    out, states = model(batch['img'].cuda(), batch['img_metas'], return_loss=False)      # the fork's (predictions, states) contract
    assert len(states) == 2 and len(states[0]['feats']) == 4 and states[0]['seg_logits'].shape[0] == 6
    pred = torch.from_numpy(np.stack(out)).long()
    assert pred.shape == ref.shape
    assert (pred != ref).float().mean() < 2e-3
    lab8, low = model.inference(batch['img'].cuda())
    assert float((low.cpu() - logits).abs().max() / logits.abs().max()) < 1e-3
    # confusion statistics exact vs numpy on identical predictions
    gt = batch['gt_semantic_seg'][:, 0]
    acc = AreaAccumulator(6)
    acc.update(lab8, gt.cuda())
    inter, union, pa, la = acc.areas()
    p, l = lab8.cpu().numpy().reshape(-1), gt.numpy().reshape(-1)
    keep = l != 255
    for c in range(6):
        assert inter[c] == np.sum((p == c) & (l == c) & keep)
        assert pa[c] == np.sum((p == c) & keep) and la[c] == np.sum((l == c) & keep)
    m = total_area_to_metrics(inter, union, pa, la)
    res = evaluate(model, [dict(img=batch['img'].cuda(), gt_semantic_seg=batch['gt_semantic_seg'].cuda())], 6)
    assert abs(res['mIoU'] - 100 * np.nanmean(m['IoU'])) < 1e-9 and 0 <= res['aAcc'] <= 100


def test_runner_checkpoint_resume(tmp_path):
    import pfst_amd  # noqa: F401
    from pfst_amd.config import Config
    from pfst_amd.data import synthetic_loader
    from pfst_amd.optim import build_optimizer
    from pfst_amd.presets import LR_CONFIG, OPTIMIZER
    from pfst_amd.registry import UDA
    from pfst_amd.runner import IterBasedRunner, find_latest_checkpoint

    def make():
        cfg = Config(dict(runner=dict(type='IterBasedRunner', max_iters=4), lr_config=dict(LR_CONFIG),
                          log_config=dict(interval=2), checkpoint_config=dict(interval=2), optimizer=dict(OPTIMIZER)))
        model = UDA.build(uda_cfg(threshold=0.3)).cuda()
        opt = build_optimizer(model, cfg.optimizer)
        return model, opt, IterBasedRunner(model, opt, cfg, str(tmp_path), log=lambda s: None)

    model, opt, runner = make()
    loader = synthetic_loader(2, 128, 6, device='cuda')
    runner.run(loader, max_iters=2)
    ck = find_latest_checkpoint(str(tmp_path))
    assert ck and ck.endswith('latest.pth') and model.local_iter == 2
    w_before = model.state_dict()['model.decode_head.conv_seg.weight'].cpu().clone()
    model2, opt2, runner2 = make()
    runner2.resume(ck)
    assert runner2.iter == 2 and model2.local_iter == 2
    assert torch.equal(model2.state_dict()['model.decode_head.conv_seg.weight'].cpu(), w_before)
    runner2.run(loader, max_iters=4)
    assert runner2.iter == 4 and model2.local_iter == 4
    st = list(opt2._flat.values())[0]
    assert st['step'] == 4            # Adam moments and step count survived the resume


def test_train_cli_on_tile_folders_with_validation_and_test_cli(tmp_path):
    """tools/train.py on folder datasets through the config's own pipelines, the validation pass every `evaluation.interval`
    iterations (apis/train.py:152-168; dead in round 1), then tools/test.py on the PFGST checkpoint with the reference's key
    revision (tools/test.py:237-242)."""
    import json
    import os
    import sys
    from PIL import Image
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import test as test_cli
    import train as train_cli
    from test_data_pipeline_cpu import NORM, SOURCE, TARGET, TEST, _tile
    for dom, n in (('pots', 4), ('vaih', 3)):
        os.makedirs(tmp_path / dom / 'img_dir/train'), os.makedirs(tmp_path / dom / 'ann_dir/train')
        for i in range(n):
            img, seg = _tile(7 * n + i, 256)
            Image.fromarray(img).save(tmp_path / dom / 'img_dir/train' / f't{i}.png')
            Image.fromarray(seg).save(tmp_path / dom / 'ann_dir/train' / f't{i}.png')
    small = lambda pl: [dict(s, crop_size=(128, 128)) if s['type'] == 'RandomCrop' else dict(s, size=(128, 128)) if s['type'] == 'Pad' else
                        dict(s, img_scale=(192, 192)) if s['type'] == 'Resize' else s for s in pl]
    test_pl = [TEST[0], dict(TEST[1], img_scale=(128, 128))]
    loader = dict(reduce_zero_label=True)
    ds = lambda dom, pl: dict(type='ISPRSDataset', data_root=str(tmp_path / dom), img_dir='img_dir/train', ann_dir='ann_dir/train',
                              gt_seg_map_loader_cfg=loader, pipeline=pl)
    from pfst_amd.presets import LR_CONFIG, OPTIMIZER
    cfg = uda_cfg(threshold=0.3)
    model_cfg = cfg.pop('model')
    cfg.pop('max_iters')
    text = ('model = %r\nuda = %r\noptimizer = %r\nlr_config = %r\nrunner = dict(type="IterBasedRunner", max_iters=2)\n'
            'checkpoint_config = dict(by_epoch=False, interval=2)\nevaluation = dict(interval=2, metric="mIoU")\nlog_config = dict(interval=1)\n'
            'seed = 0\ndata = %r\n') % (model_cfg, cfg, dict(OPTIMIZER), dict(LR_CONFIG),
                                       dict(samples_per_gpu=2, workers_per_gpu=0,
                                            train=dict(type='UDADataset', source=ds('pots', small(SOURCE)), target=ds('vaih', small(TARGET)),
                                                       rare_class_sampling=None),
                                            val=ds('vaih', test_pl), test=ds('vaih', test_pl)))
    cfg_path = tmp_path / 'toy_pfst.py'
    cfg_path.write_text(text)
    work = tmp_path / 'work'
    train_cli.main([str(cfg_path), '--work-dir', str(work), '--seed', '0'])
    lines = [json.loads(l) for l in open(work / 'log.json')]
    assert [l['iter'] for l in lines if l['mode'] == 'train'] == [1, 2]
    val = [l for l in lines if l['mode'] == 'val']
    assert len(val) == 1 and val[0]['iter'] == 2 and 0.0 <= val[0]['mIoU'] <= 100.0 and 'aAcc' in val[0]
    ck = work / 'iter_2.pth'
    assert ck.exists()
    res = test_cli.main([str(cfg_path), str(ck), '--eval', 'mIoU', '--revise-checkpoint-key', '--split', 'val'])
    assert abs(res['mIoU'] - val[0]['mIoU']) < 1e-6               # the same student weights, the same tiles
    with pytest.raises(SystemExit):
        test_cli.main([str(cfg_path), str(ck), '--split', 'val'])    # without the key revision the student's keys are missing
    # --no-validate: no validation lines
    work2 = tmp_path / 'work2'
    train_cli.main([str(cfg_path), '--work-dir', str(work2), '--seed', '0', '--no-validate'])
    assert not [l for l in map(json.loads, open(work2 / 'log.json')) if l['mode'] == 'val']


def test_inference_rescale_and_flip_follow_the_reference():
    """encoder_decoder.py:72-84,265-327: logits -> input size -> (rescale) ori_shape, two bilinear steps; a flipped input's prediction is
    flipped back.  Against torch on the oracle's eval-mode logits."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import synth_batch
    model = UDA.build(uda_cfg())
    both, student, _ = seeded_pfgst_state(O, 9)
    student = {k[6:]: v for k, v in both.items() if k.startswith('model.')}
    model.load_state_dict(both, strict=False)
    model.cuda()
    batch = synth_batch(2, 128, 6, seed=5)
    with torch.no_grad():
        feats = O.backbone_forward(student, batch['img'], train=False)
        logits, _ = O.decode_head_forward(student, feats, train=False)
        full = F.interpolate(logits, size=(128, 128), mode='bilinear', align_corners=False)
        ref = F.interpolate(full, size=(100, 144), mode='bilinear', align_corners=False).softmax(1).argmax(1)
    metas = [dict(ori_shape=(100, 144, 3), flip=False)] * 2
    lab, _ = model.get_model().inference(batch['img'].cuda(), metas, rescale=True)
    assert tuple(lab.shape) == (2, 100, 144)
    assert (lab.cpu().long() != ref).float().mean() < 2e-3
    one_step = F.interpolate(logits, size=(100, 144), mode='bilinear', align_corners=False).argmax(1)
    assert (ref != one_step).float().mean() > 0          # the two-step resize is not the one-step one: the distinction is real
    flipped = [dict(ori_shape=(100, 144, 3), flip=True, flip_direction=['horizontal', 'vertical'])] * 2
    lab_f, _ = model.get_model().inference(batch['img'].cuda(), flipped, rescale=True)
    assert torch.equal(lab_f, lab.flip(dims=(2,)).flip(dims=(1,)))
    lab_n, _ = model.get_model().inference(batch['img'].cuda(), metas, rescale=False)
    assert tuple(lab_n.shape) == (2, 128, 128)


def _eval_model_and_state(test_cfg=None):
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd.registry import UDA
    cfg = uda_cfg()
    if test_cfg is not None:
        cfg['model']['test_cfg'] = test_cfg
    model = UDA.build(cfg)
    both, _, _ = seeded_pfgst_state(O, 9)
    g = torch.Generator().manual_seed(1)
    for k, v in both.items():                      # non-trivial running statistics
        if k.endswith('running_mean'):
            v.copy_(0.05 * torch.randn(v.shape, generator=g))
        elif k.endswith('running_var'):
            v.copy_(0.8 + 0.4 * torch.rand(v.shape, generator=g))
    student = {k[6:]: v for k, v in both.items() if k.startswith('model.')}
    model.load_state_dict(both, strict=False)
    model.cuda()
    return model, student, O


@pytest.mark.parametrize('shape', [(160, 224), (96, 112)])
def test_slide_inference_follows_the_reference(shape):
    """test_cfg.mode='slide' (encoder_decoder.py:220-263): overlapping windows with the last one shifted back inside the image, crop logits
    summed and divided by the cover count, rescaled to ori_shape; an image smaller than the crop (second shape: one axis) is decoded
    without padding.  Probabilities against the oracle's line-by-line restatement, predictions through the registry API."""
    H, W = shape
    test_cfg = dict(mode='slide', crop_size=(128, 128), stride=(85, 85))
    model, student, O = _eval_model_and_state(test_cfg)
    img = torch.randn(2, 3, H, W, generator=torch.Generator().manual_seed(3))
    metas = [dict(ori_shape=(H + 16, W + 8, 3), flip=False)] * 2
    with torch.no_grad():
        ref = O.inference_probs(student, img, metas, test_cfg, True)
    seg = model.get_model()
    probs, states = seg.inference_probs(img.cuda(), metas, True)
    assert states == {} and tuple(probs.shape) == (2, 6, H + 16, W + 8)
    assert float((probs.cpu() - ref).abs().max()) < 1e-3 * float(ref.max())
    out, st = model(img.cuda(), metas, return_loss=False)
    assert st == [{}, {}]
    pred = torch.from_numpy(np.stack(out)).long()
    assert (pred != ref.argmax(1)).float().mean() < 2e-3
    # the window kernels themselves are exact: same sums as F.pad + add on identical crops
    from pfst_amd import hip_ops as ops
    a, b = torch.randn(2, 6, 40, 56, generator=torch.Generator().manual_seed(5)), torch.randn(2, 6, 40, 56, generator=torch.Generator().manual_seed(6))
    preds, count = torch.zeros(2, 6, 64, 80, device='cuda'), torch.zeros(2, 1, 64, 80, device='cuda')
    ops.window_accumulate_(preds, count, a.cuda(), 0, 0)
    ops.window_accumulate_(preds, count, b.cuda(), 24, 24)
    ops.window_accumulate_(preds, count, a.cuda(), 24, 0)
    want = F.pad(a, (0, 24, 0, 24)) + F.pad(b, (24, 0, 24, 0)) + F.pad(a, (0, 24, 24, 0))
    cnt = F.pad(torch.ones(2, 1, 40, 56), (0, 24, 0, 24)) + F.pad(torch.ones(2, 1, 40, 56), (24, 0, 24, 0)) + F.pad(torch.ones(2, 1, 40, 56), (0, 24, 24, 0))
    assert torch.equal(preds.cpu(), want) and torch.equal(count.cpu(), cnt)


@pytest.mark.parametrize('mode', ['whole', 'slide'])
def test_aug_test_averages_the_views_like_the_reference(mode):
    """aug_test (encoder_decoder.py:355-372; base.py:74-99 dispatches a list of views to it): three views -- plain, horizontally flipped,
    a larger scale flipped vertically -- each mapped back to ori_shape and un-flipped, probabilities averaged, arg-max."""
    test_cfg = dict(mode='whole') if mode == 'whole' else dict(mode='slide', crop_size=(96, 96), stride=(64, 64))
    model, student, O = _eval_model_and_state(test_cfg)
    g = torch.Generator().manual_seed(11)
    base = torch.randn(2, 3, 128, 128, generator=g)
    big = F.interpolate(base, size=(160, 160), mode='bilinear', align_corners=False)
    imgs = [base, base.flip(3), big.flip(2)]
    metas = [[dict(ori_shape=(128, 128, 3), flip=False)] * 2,
             [dict(ori_shape=(128, 128, 3), flip=True, flip_direction='horizontal')] * 2,
             [dict(ori_shape=(128, 128, 3), flip=True, flip_direction='vertical')] * 2]
    with torch.no_grad():
        ref_pred, ref_prob = O.aug_test(student, [i.clone() for i in imgs], metas, test_cfg)
    out, st = model([i.cuda() for i in imgs], metas, return_loss=False)
    assert st == {} and len(out) == 2 and out[0].shape == (128, 128)
    pred = torch.from_numpy(np.stack(out)).long()
    assert (pred != ref_pred).float().mean() < 2e-3
    seg = model.get_model()
    p0, _ = seg.inference_probs(imgs[1].cuda(), metas[1], True)
    with torch.no_grad():
        r0 = O.inference_probs(student, imgs[1], metas[1], test_cfg, True)
    assert float((p0.cpu() - r0).abs().max()) < 1e-3 * float(r0.max())
    # softmax / arg-max / flip kernels against torch on identical inputs: probabilities to 2 ulp, labels and flips exact
    from pfst_amd import hip_ops as ops
    z = torch.randn(2, 6, 33, 47, generator=g) * 3
    sm = ops.softmax_nchw(z.cuda()).cpu()
    assert float((sm - z.softmax(1)).abs().max()) < 3e-7
    assert torch.equal(ops.argmax_nchw(sm.cuda()).cpu().long(), sm.argmax(1))
    tie = torch.zeros(1, 6, 4, 4); tie[:, 2] = 1.0; tie[:, 4] = 1.0
    assert int(ops.argmax_nchw(tie.cuda()).max()) == 2 and int(ops.argmax_nchw(tie.cuda()).min()) == 2          # first maximal class
    assert torch.equal(ops.flip_planes(z.cuda(), horizontal=True).cpu(), z.flip(3)) and torch.equal(ops.flip_planes(z.cuda(), vertical=True).cpu(), z.flip(2))
    assert torch.equal(ops.div_scalar_(z.clone().cuda(), 3).cpu(), z / 3)
