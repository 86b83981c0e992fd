"""`--deterministic` (rsiseg/apis/train.py:52-68, tools/train.py:59-61 of the reference: cudnn.deterministic = True): the kernel library's
fixed-order mode (pfst_set_deterministic).  Two runs of the same train step from the same state give a BIT-IDENTICAL gradient arena, losses
and updated weights -- which the default mode does not promise (its split-K weight gradients and BatchNorm-backward reductions end in atomic
adds whose order varies) -- and the fixed-order step agrees with the default one to that summation-order noise."""
import random

import numpy as np
import pytest
import torch

from helpers import seeded_pfgst_state, to_dev, uda_cfg

pytestmark = pytest.mark.gpu


def _one_step(det, seed=123, overlap=None, early_read=None):
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd import hip_ops
    from pfst_amd.optim import build_optimizer
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import synth_batch
    from pfst_amd import layers
    hip_ops.set_deterministic(det)
    prev_overlap = (layers.WGRAD_STREAM, layers.FORK_TEACHER)
    if overlap is not None:
        layers.set_overlap(*overlap)
    prev_early = layers.EARLY_LOG_READ
    if early_read is not None:
        layers.EARLY_LOG_READ = early_read
    try:
        model = UDA.build(uda_cfg(threshold=0.30, dropout=0.1))
        both, _, _ = seeded_pfgst_state(O, 9)
        model.load_state_dict(both, strict=False)
        model.cuda()
        opt = build_optimizer(model, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01))
        batch = to_dev(synth_batch(2, 128, 6, seed=77), 'cuda')
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed); torch.cuda.manual_seed_all(seed)
        out0 = model.train_step(batch, opt)['log_vars']
        g0 = model.student_arena.grad.clone()
        out1 = model.train_step(batch, opt)['log_vars']          # second step: EMA path, AdamW state
        torch.cuda.synchronize()
        arena = model.student_arena
        layout = [(n, arena.offsets[n], int(np.prod(arena.shapes[n]))) for n in arena.names]
        return out0, g0.cpu(), out1, arena.grad.clone().cpu(), arena.data.clone().cpu(), layout
    finally:
        hip_ops.set_deterministic(False)
        layers.set_overlap(*prev_overlap)
        layers.EARLY_LOG_READ = prev_early


def test_deterministic_mode_gives_bit_identical_gradients():
    from pfst_amd import hip_ops
    assert not hip_ops.is_deterministic()
    a = _one_step(True)
    b = _one_step(True)
    assert not hip_ops.is_deterministic()
    diff = [(n, float((a[1][o:o + k] - b[1][o:o + k]).abs().max()), float((a[1][o:o + k] - b[1][o:o + k]).norm() / a[1][o:o + k].norm()), (a[1][o:o + k] != b[1][o:o + k]).nonzero().flatten().tolist()[:12], [round(float(v), 8) for v in a[1][o:o + k][a[1][o:o + k] != b[1][o:o + k]][:6]], [round(float(v), 8) for v in b[1][o:o + k][a[1][o:o + k] != b[1][o:o + k]][:6]])
            for n, o, k in a[5] if not torch.equal(a[1][o:o + k], b[1][o:o + k])]
    assert not diff, f'{len(diff)} of {len(a[5])} gradient tensors differ between two deterministic runs: {diff[:12]}'
    assert torch.equal(a[1], b[1]), 'gradient arena of the first step must be bit-identical between two deterministic runs'
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4]), 'second step (gradients and updated weights)'
    # log values: the cross-entropy / similarity loss SUMS still meet in fp64 atomics (they feed no gradient): equal to fp64 round-off
    for k in a[0]:
        assert abs(a[0][k] - b[0][k]) <= 1e-12 * max(1.0, abs(a[0][k])), k
    # ... whatever the stream schedule: with the teacher pass and the weight gradients on side streams (the product default) or everything on one
    # stream, the fixed-order step is the SAME step bit for bit -- co-running kernels must not change a single gradient element (round 5 found
    # one kernel that did: tests/test_hip_ops.py::test_depthwise_backward_is_exact_beside_a_weight_gradient_kernel)
    on, off = _one_step(True, overlap=(True, True)), _one_step(True, overlap=(False, False))
    diff = [n for n, o, k in on[5] if not torch.equal(on[1][o:o + k], off[1][o:o + k])]
    assert not diff, f'gradient tensors that depend on the stream schedule: {diff[:12]}'
    assert torch.equal(on[3], off[3]) and torch.equal(on[4], off[4])
    # and the fixed-order sums are the same mathematics as the default ones
    c = _one_step(False)
    rel = float((a[1].double() - c[1].double()).norm() / c[1].double().norm())
    print(f'   deterministic vs default gradient arena: {rel:.2e}')
    assert rel < 1e-4, rel
    for k in a[0]:
        assert abs(a[0][k] - c[0][k]) <= 1e-5 * max(abs(c[0][k]), 1e-2), (k, a[0][k], c[0][k])


def test_log_values_read_before_the_backward_sweep_are_the_steps_log_values():
    """layers.EARLY_LOG_READ (round 5, late): the packed log values are copied to the host in front of `tape.backward()` -- the reference parses
    its losses at the same point (`_parse_losses` before `total_loss.backward()`, /root/reference/rsiseg/models/uda/pfgst.py:338-344) -- so the
    step's blocking read waits for the forward passes only and the host queues the next step while the device still works on the backward sweep
    and the optimizer step.  Pure scheduling: in the fixed-order mode two consecutive steps give BIT-IDENTICAL gradients and weights with the
    copy in front of and behind the sweep, and the same 14 log values (fp64 round-off of their atomics)."""
    a, b = _one_step(True, early_read=True), _one_step(True, early_read=False)
    assert torch.equal(a[1], b[1]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    for la, lb in ((a[0], b[0]), (a[2], b[2])):
        assert list(la) == list(lb) and len(la) >= 14
        for k in la:
            assert abs(la[k] - lb[k]) <= 1e-12 * max(1.0, abs(la[k])), (k, la[k], lb[k])
