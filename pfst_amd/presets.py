"""The four shipped PFST configurations as plain dicts (values of the reference's
configs/_base_/models/deeplabv3plus_r50-d8.py:2-46, configs/_base_/uda/pfst.py:7-34,
configs/_base_/schedules/adamw_40k.py:4-21 and configs/pfst/*.py:17-53 after `_base_` merging), so that bench.py
and the GPU tests can build the workload without the reference tree.  `Config.fromfile` loads the reference's own
files unchanged when they are available (tests/test_config_dropin.py)."""
import copy

NORM_CFG = dict(type='BN', requires_grad=True)


def model_cfg(num_classes=6, in_channels=3, dropout=0.1):
    return dict(
        type='EncoderDecoder', pretrained=None,
        backbone=dict(type='ResNetV1c', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), dilations=(1, 1, 2, 4),
                      strides=(1, 2, 1, 1), norm_cfg=dict(NORM_CFG), norm_eval=False, style='pytorch',
                      contract_dilation=True, in_channels=in_channels),
        decode_head=dict(type='DepthwiseSeparableASPPHead', in_channels=2048, in_index=3, channels=512,
                         dilations=(1, 12, 24, 36), c1_in_channels=256, c1_channels=48, dropout_ratio=dropout,
                         num_classes=num_classes, norm_cfg=dict(NORM_CFG), align_corners=False,
                         loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)),
        auxiliary_head=dict(type='FCNHead', in_channels=1024, in_index=2, channels=256, num_convs=1, concat_input=False,
                            dropout_ratio=dropout, num_classes=num_classes, norm_cfg=dict(NORM_CFG), align_corners=False,
                            loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.4)),
        train_cfg=dict(), test_cfg=dict(mode='whole'))


def uda_cfg(num_classes=6, in_channels=3, dropout=0.1, blur=True, color_jitter_probability=0.2, downscale=0.5,
            pseudo_threshold=0.98, max_iters=40000):
    return dict(
        type='PFGST', alpha=0.999, pseudo_threshold=pseudo_threshold, pseudo_weight_ignore_top=0,
        pseudo_weight_ignore_bottom=0, imnet_feature_dist_lambda=0, imnet_feature_dist_classes=None,
        imnet_feature_dist_scale_min_ratio=None, mix='class', blur=blur, color_jitter_strength=0.2,
        color_jitter_probability=color_jitter_probability, print_grad_magnitude=False, thre_type='all',
        trg_loss_weight=1., use_decoded_feats=True,
        aux_losses=[dict(type='PFGSTLoss', kernel_size=3, dilation=2, top_k=3,
                         weights={'src_pos': 0.1, 'src_neg': 0.1, 'sim_pos': 0.1, 'sim_neg': 0.1,
                                  'src_pos_std': 0.1, 'src_neg_std': 0.1},
                         sim_type='cosine', feat_level=None, detach_unfold=True, downscale=downscale)],
        model=model_cfg(num_classes, in_channels, dropout), max_iters=max_iters)


OPTIMIZER = dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
LR_CONFIG = dict(policy='poly', warmup='linear', warmup_iters=1500, warmup_ratio=1e-6, power=1.0, min_lr=0.0, by_epoch=False)

# BASELINE.json configs[1..4]: name -> (num_classes, in_channels, tile size, per-GPU batch, PFGSTLoss downscale)
WORKLOADS = {
    'pfst_pots_irrg2vaih_irrg_deeplabv3plus_r50-d8': dict(num_classes=6, in_channels=3, size=1024, per_gpu_batch=8, downscale=0.5),
    'pfst_vaih_irrg2pots_irrg_deeplabv3plus_r50-d8': dict(num_classes=6, in_channels=3, size=1024, per_gpu_batch=8, downscale=0.5),
    'pfst_inria_da_deeplabv3plus_r50-d8': dict(num_classes=2, in_channels=3, size=1024, per_gpu_batch=8, downscale=0.5),
    'pfst_season_net_sp2fa_deeplabv3plus_r50-d8': dict(num_classes=33, in_channels=10, size=512, per_gpu_batch=8, downscale=1,
                                                       strong_aug_denorm_type='none'),
}


def workload_cfg(name, **overrides):
    w = dict(WORKLOADS[name])
    cfg = uda_cfg(w['num_classes'], w['in_channels'], downscale=w['downscale'])
    if 'strong_aug_denorm_type' in w:                   # configs/pfst/pfst_season_net_sp2fa_*.py:53
        cfg['strong_aug_denorm_type'] = w['strong_aug_denorm_type']
    in_channels = overrides.pop('in_channels', None)    # e.g. the SHIPPED 3-band SeasonNet variant of BASELINE config #5
    if in_channels is not None:
        cfg['model']['backbone']['in_channels'] = w['in_channels'] = in_channels
    cfg.update(overrides)
    return copy.deepcopy(cfg), w
