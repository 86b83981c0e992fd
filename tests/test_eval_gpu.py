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
