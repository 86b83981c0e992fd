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
