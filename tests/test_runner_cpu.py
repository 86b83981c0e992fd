"""CPU tests of the trainer glue: LR schedule values, CLI surface, checkpoint key revision."""
import math

from pfst_amd.optim import poly_lr


def test_poly_lr_with_linear_warmup_matches_mmcv_formulas():
    base, mx = 6e-5, 40000
    # mmcv PolyLrUpdaterHook: (base - min_lr) * (1 - it/max)^power + min_lr ; linear warm-up: lr * (1 - (1 - it/wi) * (1 - ratio))
    assert poly_lr(base, 0, mx) == base * (1 - (1 - 1e-6))
    it = 750
    regular = base * (1 - it / mx)
    assert math.isclose(poly_lr(base, it, mx), regular * (1 - (1 - it / 1500) * (1 - 1e-6)), rel_tol=1e-12)
    assert math.isclose(poly_lr(base, 1500, mx), base * (1 - 1500 / mx), rel_tol=1e-12)
    assert math.isclose(poly_lr(base, 39999, mx), base * (1 / mx), rel_tol=1e-9)


def test_train_cli_flags_of_the_reference_parse():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('train_cli', os.path.join(os.path.dirname(__file__), '..', 'tools', 'train.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    a = m.parse_args(['cfg.py', '--work-dir', 'w', '--seed', '3', '--deterministic', '--cfg-options', 'a.b=1', 'c=[1,2]',
                      '--launcher', 'pytorch', '--local_rank', '2', '--auto-resume', '--no-validate', '--gpu-id', '1'])
    assert a.config == 'cfg.py' and a.work_dir == 'w' and a.seed == 3 and a.launcher == 'pytorch' and a.local_rank == 2
    from pfst_amd.config import parse_cfg_options
    assert parse_cfg_options(a.cfg_options) == {'a.b': 1, 'c': [1, 2]}
    assert a.deterministic is True          # switches the kernel library to fixed-order sums (tests/test_deterministic_gpu.py)


def test_checkpoint_roundtrip_keeps_reference_layout_and_local_iter(tmp_path):
    import torch
    import pfst_amd  # noqa: F401
    from pfst_amd.presets import uda_cfg
    from pfst_amd.registry import UDA
    m = UDA.build(uda_cfg())
    m.local_iter = 17
    sd = m.state_dict()
    assert sd['_extra_state'] == {'local_iter': 17}
    torch.save(sd, tmp_path / 'c.pth')
    m2 = UDA.build(uda_cfg())
    m2.load_state_dict(torch.load(tmp_path / 'c.pth', weights_only=False))
    assert m2.local_iter == 17
    # tools/test.py:237-242 style: strip 'model.' to load the student into a bare EncoderDecoder
    from pfst_amd.registry import build_segmentor
    seg = build_segmentor(uda_cfg()['model'])
    student = {k[len('model.'):]: v for k, v in sd.items() if k.startswith('model.')}
    assert seg.load_state_dict(student, strict=True)
