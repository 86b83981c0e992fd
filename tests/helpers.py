"""Shared builders for the tests: the PFST config dict (values of configs/pfst/pfst_pots_irrg2vaih_irrg_*.py
merged with its _base_ files) and state_dict helpers."""
from collections import OrderedDict

import torch


def model_cfg(num_classes=6, in_channels=3, dropout=0.0):
    norm_cfg = dict(type='BN', requires_grad=True)
    return dict(
        type='EncoderDecoder', pretrained=None,
        backbone=dict(type='ResNetV1c', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), dilations=(1, 1, 2, 4),
                      strides=(1, 2, 1, 1), norm_cfg=norm_cfg, norm_eval=False, style='pytorch', contract_dilation=True,
                      in_channels=in_channels),
        decode_head=dict(type='DepthwiseSeparableASPPHead', in_channels=2048, in_index=3, channels=512,
                         dilations=(1, 12, 24, 36), c1_in_channels=256, c1_channels=48, dropout_ratio=dropout,
                         num_classes=num_classes, norm_cfg=norm_cfg, align_corners=False,
                         loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)),
        auxiliary_head=dict(type='FCNHead', in_channels=1024, in_index=2, channels=256, num_convs=1, concat_input=False,
                            dropout_ratio=dropout, num_classes=num_classes, norm_cfg=norm_cfg, align_corners=False,
                            loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.4)),
        train_cfg=dict(), test_cfg=dict(mode='whole'))


def uda_cfg(num_classes=6, in_channels=3, dropout=0.0, threshold=0.98, blur=False, jitter_p=2.0):
    return dict(
        type='PFGST', alpha=0.999, pseudo_threshold=threshold, pseudo_weight_ignore_top=0, pseudo_weight_ignore_bottom=0,
        imnet_feature_dist_lambda=0, imnet_feature_dist_classes=None, imnet_feature_dist_scale_min_ratio=None,
        mix='class', blur=blur, color_jitter_strength=0.2, color_jitter_probability=jitter_p, print_grad_magnitude=False,
        thre_type='all', trg_loss_weight=1., use_decoded_feats=True,
        aux_losses=[dict(type='PFGSTLoss', kernel_size=3, dilation=2, top_k=3,
                         weights={'src_pos': 0.1, 'src_neg': 0.1, 'sim_pos': 0.1, 'sim_neg': 0.1,
                                  'src_pos_std': 0.1, 'src_neg_std': 0.1},
                         sim_type='cosine', feat_level=None, detach_unfold=True, downscale=0.5)],
        model=model_cfg(num_classes, in_channels, dropout), max_iters=40000)


def seeded_pfgst_state(oracle_mod, seed, num_classes=6, in_channels=3):
    """(full 848-key state_dict, student dict, teacher dict) filled like tests/golden/make_golden.py does."""
    from pfst_amd.synthetic import fill_state_dict
    base = oracle_mod.init_state_dict(num_classes, in_channels)
    both = OrderedDict(('model.' + k, v.clone()) for k, v in base.items())
    both.update(('ema_model.' + k, v.clone()) for k, v in base.items())
    fill_state_dict(both, seed)
    student = OrderedDict((k[6:], v) for k, v in both.items() if k.startswith('model.'))
    teacher = OrderedDict((k[10:], v) for k, v in both.items() if k.startswith('ema_model.'))
    return both, student, teacher


def to_dev(batch, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}


def assert_live_target_side(log, ex=None, min_frac=0.15):
    """A whole-step parity input must exercise the target-side half of PFGSTLoss (pfgst_loss.py:62-71,203-234): a valid region
    (source label != 255 AND all nine dilated neighbours un-mixed) of at least `min_frac` of the loss grid, so that loss_sim_pos/neg
    and their gradient into the mixed-pass logits are non-zero (with <= 1 valid pixel the reference returns zeros(1)).
    log: the step's log_vars; ex: the oracle's extras (holds the mask) when the test ran the oracle."""
    assert log['loss_sim_pos'] != 0.0 and log['loss_sim_neg'] != 0.0, ('dead target side', log['loss_sim_pos'], log['loss_sim_neg'])
    if ex is not None:
        frac = float(ex['mask'].float().mean())
        assert frac >= min_frac, f'valid target region is {frac:.3f} of the grid'
