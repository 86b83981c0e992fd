"""The remaining known-answer / structural tests the reference's own test-suite holds for this path,
restated against the oracle and the host-side mirror (CPU only).  CE / accuracy known answers live in
test_oracle_golden.py; here: weighted reduction, contract_dilation, DW-ASPP shapes, mIoU vs the legacy confusion matrix."""
import numpy as np
import pytest
import torch

from oracle import pfst_oracle as O


def test_weight_reduce_loss():
    # tests/test_models/test_losses/test_utils.py:9-41 : weighted 'mean' = (loss * weight).mean() over ALL elements
    g = torch.Generator().manual_seed(0)
    logits = torch.rand(1, 3, 4, 4, generator=g)
    label = torch.randint(0, 3, (1, 4, 4), generator=g)
    weight = torch.zeros(1, 4, 4)
    weight[:, :2, :2] = 1
    per_px = torch.nn.functional.cross_entropy(logits, label, reduction='none')
    assert float(O.ce_loss(logits, label, weight=weight)) == pytest.approx(float((per_px * weight).mean()), rel=1e-6)
    assert float(O.ce_loss(logits, label)) == pytest.approx(float(per_px.mean()), rel=1e-6)
    # loss_weight scales the reduced value (cross_entropy_loss.py:290-297)
    assert float(O.ce_loss(logits, label, loss_weight=0.4)) == pytest.approx(0.4 * float(per_px.mean()), rel=1e-6)


def test_contract_dilation_structure():
    # tests/test_models/test_backbones/test_resnet.py:250-266 : the first block of a dilated stage uses dilation // 2
    import pfst_amd  # noqa: F401
    from pfst_amd.models import ResNetV1c
    net = ResNetV1c(depth=50, strides=(1, 2, 1, 1), dilations=(1, 1, 2, 4), contract_dilation=True)
    assert net.layer3[0].conv2.dilation == 1 and net.layer3[0].conv2.padding == 1
    assert all(b.conv2.dilation == 2 and b.conv2.padding == 2 for b in list(net.layer3)[1:])
    assert net.layer4[0].conv2.dilation == 2
    assert all(b.conv2.dilation == 4 for b in list(net.layer4)[1:])
    plain = ResNetV1c(depth=50, strides=(1, 2, 1, 1), dilations=(1, 1, 2, 4), contract_dilation=False)
    assert plain.layer3[0].conv2.dilation == 2 and plain.layer4[0].conv2.dilation == 4
    # the oracle's functional backbone follows the same rule: stride-8 outputs, 4 stages
    sd = O.init_state_dict(6, 3, seed=1)
    with torch.no_grad():
        outs = O.backbone_forward(sd, torch.randn(1, 3, 64, 64), train=True)
    assert [tuple(o.shape[1:]) for o in outs] == [(256, 16, 16), (512, 8, 8), (1024, 8, 8), (2048, 8, 8)]


def test_dw_aspp_head_shapes_and_dilations():
    # tests/test_models/test_heads/test_aspp_head.py:39-76 : output has the c1 resolution (odd sizes), atrous rates on the
    # depthwise convs, plain 1x1 for rate 1
    import pfst_amd  # noqa: F401
    from pfst_amd.models import DepthwiseSeparableASPPHead
    head = DepthwiseSeparableASPPHead(c1_in_channels=4, c1_channels=2, in_channels=16, channels=8, num_classes=19,
                                      dilations=(1, 12, 24), in_index=3, norm_cfg=dict(type='BN', requires_grad=True))
    assert head.c1_bottleneck.conv.cin == 4 and head.c1_bottleneck.conv.cout == 2
    assert head.aspp_modules[0].conv.dilation == 1 and head.aspp_modules[0].conv.k == 1
    assert head.aspp_modules[1].depthwise_conv.conv.dilation == 12
    assert head.aspp_modules[2].depthwise_conv.conv.dilation == 24
    sd = O.init_state_dict(19, 3, seed=2)
    feats = (torch.randn(2, 256, 45, 45), None, None, torch.randn(2, 2048, 21, 21))
    with torch.no_grad():
        logits, features = O.decode_head_forward(sd, feats, train=True)
    assert logits.shape == (2, 19, 45, 45) and features.shape == (2, 512, 21, 21)


def _legacy_mean_iou(results, labels, num_classes, ignore_index):
    """tests/test_metrics.py:9-44 (confusion-matrix formulation), restated"""
    tot = np.zeros((num_classes, num_classes), dtype=np.float64)
    for r, l in zip(results, labels):
        keep = l != ignore_index
        tot += np.bincount(num_classes * l[keep] + r[keep], minlength=num_classes ** 2).reshape(num_classes, num_classes)
    with np.errstate(divide='ignore', invalid='ignore'):
        return (np.diag(tot).sum() / tot.sum(), np.diag(tot) / tot.sum(1), np.diag(tot) / (tot.sum(1) + tot.sum(0) - np.diag(tot)),
                2 * np.diag(tot) / (tot.sum(1) + tot.sum(0)))


def _areas(results, labels, n, ignore_index):
    inter, pa, la = np.zeros(n), np.zeros(n), np.zeros(n)
    for r, l in zip(results, labels):
        keep = l != ignore_index
        r, l = r[keep], l[keep]
        inter += np.bincount(r[r == l], minlength=n)[:n]
        pa += np.bincount(r, minlength=n)[:n]
        la += np.bincount(l, minlength=n)[:n]
    return inter, pa + la - inter, pa, la


def test_mean_iou_and_dice_match_the_legacy_confusion_matrix():
    # tests/test_metrics.py:219-243 (mIoU) and :246-268 (mDice), incl. nan_to_num for absent classes
    import pfst_amd  # noqa: F401
    from pfst_amd.evaluation import total_area_to_metrics
    rng = np.random.RandomState(0)
    n, ign = 19, 255
    results = rng.randint(0, n, size=(10, 30, 30))
    label = rng.randint(0, n, size=(10, 30, 30))
    label[:, 2, 5:10] = ign
    m = total_area_to_metrics(*_areas(results, label, n, ign), metrics=('mIoU', 'mDice'))
    all_acc, acc, iou, dice = _legacy_mean_iou(results, label, n, ign)
    assert m['aAcc'] == all_acc
    assert np.allclose(m['Acc'], acc) and np.allclose(m['IoU'], iou) and np.allclose(m['Dice'], dice)
    results = rng.randint(0, 5, size=(10, 30, 30))
    label = rng.randint(0, 4, size=(10, 30, 30))
    m = total_area_to_metrics(*_areas(results, label, n, ign), nan_to_num=-1)
    assert m['Acc'][-1] == -1 and m['IoU'][-1] == -1
    # every class present once (test_metrics.py:210-217): no NaN
    label = np.arange(59)[None]
    m = total_area_to_metrics(*_areas(label.copy(), label, 59, ign))
    assert not np.any(np.isnan(m['IoU']))


def test_fscore_definition():
    # rsiseg/core/evaluation/metrics.py:8-23 f_score(precision, recall, beta)
    import pfst_amd  # noqa: F401
    from pfst_amd.evaluation import total_area_to_metrics
    inter, pa, la = np.array([6., 2.]), np.array([8., 4.]), np.array([10., 2.])
    m = total_area_to_metrics(inter, pa + la - inter, pa, la, metrics=('mFscore',), beta=2)
    p, r = inter / pa, inter / la
    assert np.allclose(m['Fscore'], (1 + 4) * p * r / (4 * p + r))
    assert np.allclose(m['Precision'], p) and np.allclose(m['Recall'], r)


def test_class_weight_files(tmp_path):
    # rsiseg/models/losses/utils.py:10-25 get_class_weight: list | .npy | whatever mmcv.load reads (json / yaml / pkl)
    import json
    import pickle
    import pfst_amd  # noqa: F401
    from pfst_amd.models import CrossEntropyLoss, get_class_weight
    w = [0.5, 1.0, 1.5, 2.0, 0.7, 1.2]
    np.save(tmp_path / 'w.npy', np.array(w))
    (tmp_path / 'w.json').write_text(json.dumps(w))
    (tmp_path / 'w.yaml').write_text('\n'.join(f'- {v}' for v in w))
    (tmp_path / 'w.pkl').write_bytes(pickle.dumps(w))
    assert get_class_weight(w) is w and get_class_weight(None) is None
    for name in ('w.npy', 'w.json', 'w.yaml', 'w.pkl'):
        assert np.allclose(get_class_weight(str(tmp_path / name)), w), name
    assert np.allclose(CrossEntropyLoss(class_weight=str(tmp_path / 'w.json')).class_weight, w)
    with pytest.raises(TypeError):
        get_class_weight(str(tmp_path / 'w.txt'))
