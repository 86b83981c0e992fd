"""Drop-in boundary on the CPU: the reference's own config files load unchanged and build the registered types with
the reference's checkpoint layout.  (The config files exist only in the build container; skipped on the GPU box.)"""
import os

import numpy as np
import pytest

REF_CFG = '/root/reference/configs/pfst'
NAMES = ['pfst_pots_irrg2vaih_irrg', 'pfst_vaih_irrg2pots_irrg', 'pfst_inria_da', 'pfst_season_net_sp2fa']


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference configs not present')
@pytest.mark.parametrize('name', NAMES)
def test_reference_configs_load_and_build(name, golden_dir):
    import pfst_amd  # noqa: F401
    from pfst_amd.config import Config, parse_cfg_options
    from pfst_amd.registry import build_train_model
    cfg = Config.fromfile(os.path.join(REF_CFG, f'{name}_deeplabv3plus_r50-d8.py'))
    cfg.merge_from_dict(parse_cfg_options(['model.pretrained=None', 'runner.max_iters=123']))
    assert cfg.uda.type == 'PFGST' and cfg.uda.aux_losses[0]['type'] == 'PFGSTLoss'
    model = build_train_model(cfg)
    assert model.max_iters == 123
    keys = [k for k in model.state_dict().keys() if k != '_extra_state']
    gold = [str(k) for k in np.load(os.path.join(golden_dir, 'train_step.npz'))['keys']]
    assert keys == gold                              # the reference's 848 keys in the reference's order
    nc = cfg.model.decode_head.num_classes
    assert model.model.decode_head.conv_seg.weight.shape[0] == nc
    assert not any(p.requires_grad for p in model.ema_model.parameters())


def test_presets_match_reference_configs_when_available():
    from pfst_amd.presets import WORKLOADS, workload_cfg
    for name, w in WORKLOADS.items():
        cfg, _ = workload_cfg(name)
        assert cfg['model']['decode_head']['num_classes'] == w['num_classes']
        path = os.path.join(REF_CFG, name + '.py')
        if os.path.exists(path):
            from pfst_amd.config import Config
            ref = Config.fromfile(path)
            assert ref.uda.pseudo_threshold == cfg['pseudo_threshold'] and ref.uda.alpha == cfg['alpha']
            assert ref.uda.aux_losses[0]['downscale'] == w['downscale']
            assert ref.model.decode_head.num_classes == w['num_classes']


def test_unknown_options_fail_loudly():
    import pfst_amd  # noqa: F401
    from pfst_amd.presets import uda_cfg
    from pfst_amd.registry import UDA
    cfg = uda_cfg()
    cfg['imnet_feature_dist_lambda'] = 0.005          # DAFormer feature distance: not part of the PFST configs
    with pytest.raises(NotImplementedError):
        UDA.build(cfg)
    with pytest.raises(KeyError):
        UDA.build(dict(type='DACS'))


def test_winograd_dispatch_rules():
    """Host-side dispatch of the wide stride-1 3x3 layers to the Winograd path (pfst_amd/layers.py): geometry and size rules."""
    import pfst_amd  # noqa: F401
    from pfst_amd import layers
    from pfst_amd.layers import Conv2dP
    assert layers.WINOGRAD                                            # on by default (PFST_WINOGRAD=0 disables it)
    ok = Conv2dP(512, 512, 3, 1, 4, 4)                                # layer4 conv2: same-size dilated 3x3
    assert ok._wino_eligible()
    assert Conv2dP(2560, 512, 3, 1, 1, 1)._wino_eligible()            # head bottleneck
    assert Conv2dP(256, 256, 3, 1, 2, 2)._wino_eligible()             # layer3 conv2
    assert layers.WINO_MIN_CC <= layers.WINO_MIN_CC_WGRAD <= 512 * 512
    assert Conv2dP(128, 128, 3, 1, 1, 1)._wino_eligible() == (layers.ops.WINO_TILE == 4)   # layer2 conv2: pays off with F(4x4) only
    assert not Conv2dP(64, 64, 3, 1, 1, 1)._wino_eligible()           # layer1 conv2, transform-bound: stays direct
    assert not Conv2dP(512, 512, 3, 2, 1, 1)._wino_eligible()         # stride 2
    assert not Conv2dP(512, 512, 1)._wino_eligible()                  # 1x1
    assert not Conv2dP(512, 512, 3, 1, 0, 1)._wino_eligible()         # not a 'same' convolution
    assert not Conv2dP(2048, 2048, 3, 1, 12, 12, groups=2048)._wino_eligible()   # depthwise
    assert not Conv2dP(520, 512, 3, 1, 1, 1)._wino_eligible()         # Cin not a multiple of 16: generic kernel
