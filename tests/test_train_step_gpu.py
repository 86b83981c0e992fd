"""End-to-end parity of the HIP PFGST.train_step (through the registry API + C ABI):
 (1) against the golden vectors the REFERENCE produced (tests/golden/train_step.npz: two full iterations),
 (2) against the CPU oracle on the same seeded inputs: bit-exact pseudo-label / mixed-label maps,
     logits / losses / gradients within 1e-3 relative (north_star tolerance, fp32)."""
import os
import random

import numpy as np
import pytest
import torch

from helpers import seeded_pfgst_state, to_dev, uda_cfg

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
TOL = 1e-3


def _build(threshold):
    import pfst_amd  # noqa: F401
    from pfst_amd.optim import build_optimizer
    from pfst_amd.registry import UDA
    from oracle import pfst_oracle as O
    model = UDA.build(uda_cfg(threshold=threshold))
    both, student, teacher = seeded_pfgst_state(O, 9)
    missing = model.load_state_dict(both, strict=False)
    assert not missing.unexpected_keys and missing.missing_keys in ([], ['_extra_state'])
    model.cuda()
    opt = build_optimizer(model, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01))
    return model, opt, student, teacher


def rel(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_two_train_steps_match_reference_golden_and_oracle():
    from oracle import pfst_oracle as O
    from pfst_amd.synthetic import synth_batch
    gold = np.load(os.path.join(G, 'train_step.npz'))
    model, opt, student, teacher = _build(0.30)
    oracle = O.OraclePFGST(student, pseudo_threshold=0.30, teacher_sd=teacher)
    torch.set_num_threads(os.cpu_count() or 8)
    for it in range(2):
        batch = synth_batch(2, 128, 6, seed=1234 + it)
        # --- HIP product (consumes the global python/numpy RNG exactly like the reference)
        random.seed(100 + it); np.random.seed(100 + it)
        model.debug = {}
        out = model.train_step(to_dev(batch, 'cuda'), opt)
        dbg = model.debug
        # --- oracle with the same RNG stream
        random.seed(100 + it); np.random.seed(100 + it)
        olog, ex = oracle.train_step(batch, return_extras=True)
        lv = out['log_vars']
        assert list(lv.keys()) == list(olog.keys())
        assert out['num_samples'] == 2
        # bit-exact integer maps
        assert torch.equal(dbg['pseudo_label'].cpu(), ex['pseudo_label']), 'pseudo-label map'
        assert torch.equal(dbg['mix_masks'].cpu().long(), ex['masks']), 'class-mix masks'
        assert torch.equal(dbg['mixed_lbl'].cpu(), ex['mixed_lbl']), 'mixed label map'
        assert abs(int(dbg['conf_count'].item()) - ex['n_conf']) <= 2
        # fp32 tensors within 1e-3 relative
        assert rel(dbg['src_logits'], ex['src_logits']) < TOL
        assert rel(dbg['mix_logits'], ex['mix_logits']) < TOL
        assert rel(dbg['ema_dec'], ex['ema_dec']) < TOL
        assert rel(dbg['mixed_w'], ex['mixed_w']) < 1e-5
        for k in olog:
            assert abs(lv[k] - olog[k]) <= TOL * max(abs(olog[k]), 1e-2), (it, k, lv[k], olog[k])
        if it == 0:
            arena = model.student_arena
            worst = 0.0
            for name, g in ex['grads'].items():
                mine = arena.view(arena.grad, name)
                r = rel(mine, g)
                worst = max(worst, r)
                assert r < 5 * TOL, (name, r)      # per-tensor; the flat gradient is checked at 1e-3 below
            flat_o = torch.cat([g.flatten() for g in ex['grads'].values()])
            flat_m = torch.cat([arena.view(arena.grad, n).flatten() for n in ex['grads']])
            assert rel(flat_m, flat_o) < TOL
    # --- against the reference's own numbers (same seeds as make_golden.py: python/numpy seed 0 BEFORE step 0)
    model, opt, student, teacher = _build(0.30)
    random.seed(0); np.random.seed(0)
    for it in range(2):
        batch = synth_batch(2, 128, 6, seed=1234 + it)
        model.debug = {}
        out = model.train_step(to_dev(batch, 'cuda'), opt)
        keys = [str(k) for k in gold[f'it{it}_log_keys']]
        vals = gold[f'it{it}_log_vals']
        assert list(out['log_vars'].keys()) == keys
        for k, v in zip(keys, vals):
            assert abs(out['log_vars'][k] - v) <= 2 * TOL * max(abs(v), 1e-2), (it, k, out['log_vars'][k], v)
        ml = model.debug['mixed_lbl'].cpu()
        assert np.array_equal(ml.numpy(), gold[f'it{it}_mixed_lbl'])
        if it == 0:
            arena = model.student_arena
            g = arena.view(arena.grad, 'decode_head.conv_seg.weight')
            assert rel(g, torch.from_numpy(gold['it0_grad|decode_head.conv_seg.weight'])) < 2 * TOL
            g = arena.view(arena.grad, 'backbone.stem.0.weight')
            assert rel(g, torch.from_numpy(gold['it0_grad|backbone.stem.0.weight'])) < 5 * TOL
    sd = model.state_dict()
    for k in gold.files:
        if k.startswith('final|'):
            got = sd[k[6:]].detach().cpu().double().flatten()[:4096].numpy()
            assert np.allclose(got, gold[k], rtol=1e-3, atol=2.5e-4), k     # AdamW sign-like first steps, see oracle test
