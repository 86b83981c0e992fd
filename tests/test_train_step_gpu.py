"""End-to-end parity of the HIP PFGST.train_step (through the registry API + C ABI):
 (1) against the golden vectors the REFERENCE produced (tests/golden/train_step.npz: two full iterations),
 (2) against the CPU oracle on the same seeded inputs: bit-exact pseudo-label / mixed-label maps,
     logits / losses / gradients within 1e-3 relative (north_star tolerance, fp32)."""
import os
import random

import numpy as np
import pytest
import torch

from helpers import assert_live_target_side, seeded_pfgst_state, to_dev, uda_cfg

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


def assert_elementwise(a, ref, what, rtol=TOL, atol_rel=TOL):
    """element-wise mixed bound |a - ref| <= atol_rel * max|ref| + rtol * |ref| (beside the norm-wise `rel`): EVERY element of the
    end-to-end tensors (60 fp32 layers deep; logits of the N(0, .01)-initialised classifier pass through zero) within 1e-3 of the
    tensor's scale plus 1e-3 of its own value.  Measured worst element: 4.5e-4 of max|logits| (the per-link bound of
    tests/test_layer_backward_gpu.py is ten times tighter: atol 1e-4)."""
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    bound = atol_rel * float(ref.abs().max()) + rtol * ref.abs()
    worst = float(((a - ref).abs() / bound).max())
    assert worst <= 1.0, f'{what}: worst element at {worst:.2f}x the bound'


@pytest.fixture(params=['f32', 'bf16x6', 'f16x3'])
def conv_math(request):
    """run the end-to-end parity under every convolution arithmetic (fp32-input MFMA; the fp32-faithful bf16x6 and f16x3 splits)"""
    from pfst_amd import layers
    prev, layers.CONV_MATH = layers.CONV_MATH, request.param
    yield request.param
    layers.CONV_MATH = prev


def test_two_train_steps_match_reference_golden_and_oracle(conv_math):
    from oracle import pfst_oracle as O
    from pfst_amd.synthetic import synth_batch
    gold = np.load(os.path.join(G, 'train_step.npz'))
    model, opt, student, teacher = _build(0.30)
    oracle = O.OraclePFGST(student, pseudo_threshold=0.30, teacher_sd=teacher)
    from pfst_amd.hostinfo import usable_cpus
    torch.set_num_threads(usable_cpus())
    for it in range(2):
        batch = synth_batch(2, 128, 6, seed=1234 + it)
        if it > 0:
            # Start every compared step from IDENTICAL state: AdamW's first steps move each weight by ~lr*sign(g), so
            # the 2-3 % fp32 gradient noise (see below) turns into O(10 %) logit differences after one update of this
            # random-init network.  The optimiser arithmetic itself is pinned in test_hip_ops.py::test_ema_adamw_flat.
            from collections import OrderedDict
            sync = OrderedDict(('model.' + k, v.detach()) for k, v in oracle.student.items())
            sync.update(('ema_model.' + k, v.detach()) for k, v in oracle.teacher.items())
            model.load_state_dict(sync, strict=False)
        # --- oracle (fp32 CPU restatement of the reference), consuming the global python/numpy RNG like the reference
        random.seed(101 + it); np.random.seed(101 + it)
        olog, ex = oracle.train_step(batch, return_extras=True)
        # --- HIP product with the same RNG stream.  Its own pseudo-label map is checked below; the student passes
        # then use the oracle's map so that gradient parity is not polluted by arg-max near-ties (see below).
        random.seed(101 + it); np.random.seed(101 + it)
        model.debug = {}
        model.injected_pseudo = (ex['pseudo_label'].to(torch.uint8).cuda(), torch.tensor([ex['n_conf']], dtype=torch.int64).cuda())
        out = model.train_step(to_dev(batch, 'cuda'), opt)
        dbg = model.debug
        lv = out['log_vars']
        assert list(lv.keys()) == list(olog.keys())
        assert out['num_samples'] == 2
        # bit-exact integer maps
        # The label KERNEL is bit-exact on identical logits (test_hip_ops.py::test_pseudo_label_bit_exact and the
        # check right below).  End to end the teacher logits themselves differ by fp32 rounding (~1e-6), so pixels
        # whose two best classes tie to within that noise may flip: only a mismatch RATE is meaningful here.
        from pfst_amd import hip_ops
        pl_same = dbg['own_pseudo_label'].cpu() == ex['pseudo_label']
        mism = 1.0 - pl_same.float().mean().item()
        print(f'it{it}: end-to-end pseudo-label mismatch rate {mism:.2e}')
        # it 1: the weights have taken one AdamW step (~lr*sign(g), which amplifies gradient noise), random-init
        # logits are nearly flat, so the end-to-end flip rate is no longer informative -- the kernel check below is.
        assert mism < 2e-3, mism
        l64, _, _ = hip_ops.pseudo_label(ex['ema_logits_low'].cuda(), (128, 128), 0.30)
        assert torch.equal(l64.cpu(), ex['pseudo_label']), 'pseudo-label kernel must be bit exact on identical logits'
        assert torch.equal(dbg['mix_masks'].cpu().long(), ex['masks']), 'class-mix masks'
        ml_same = dbg['mixed_lbl'].cpu() == ex['mixed_lbl']
        assert bool(ml_same.all()), 'mixed label map'
        assert abs(int(dbg['own_conf_count'].item()) - ex['n_conf']) <= 4
        # fp32 tensors within 1e-3 relative
        assert rel(dbg['src_logits'], ex['src_logits']) < TOL
        assert rel(dbg['mix_logits'], ex['mix_logits']) < TOL
        assert rel(dbg['ema_dec'], ex['ema_dec']) < TOL
        assert_elementwise(dbg['src_logits'], ex['src_logits'], 'source logits')
        assert_elementwise(dbg['mix_logits'], ex['mix_logits'], 'mixed-pass logits')
        assert_elementwise(dbg['ema_logits'], ex['ema_logits_low'], 'teacher logits')
        assert_elementwise(dbg['ema_dec'], ex['ema_dec'], 'teacher decoded features')
        assert rel(dbg['mixed_w'], ex['mixed_w']) < 1e-5
        for k in olog:
            # acc_seg counts arg-max hits of nearly flat random-init logits: after the AdamW step (it 1, see above) a handful of near-tie
            # pixels of the 32768 may flip -> an absolute slack of 10 pixels there (measured: 2), as in the option-variant tests below
            tol = 100.0 * 10 / (2 * 128 * 128) if (it == 1 and k.endswith('acc_seg')) else TOL * max(abs(olog[k]), 1e-2)
            assert abs(lv[k] - olog[k]) <= tol, (it, k, lv[k], olog[k])
        assert_live_target_side(olog, ex)
        assert_live_target_side(lv)
        if it == 0:
            # Gradients.  With random-init weights the backward pass through ~70 train-mode BN layers is badly
            # conditioned: the reference's OWN fp32 CPU path differs from exact (fp64) arithmetic by 2-5 % per tensor
            # (measured below), so "1e-3 relative to the fp32 reference" cannot be met by any fp32 implementation,
            # the reference included.  The criterion used: the HIP gradient must be as close to the fp64 gradient as
            # the reference's fp32 path is (x5 slack for different-but-equally-valid fp32 summation orders, 1e-3 floor);
            # tensors next to the loss, where conditioning is fine, must meet 1e-3 directly.
            arena = model.student_arena
            o64 = O.OraclePFGST({k: (v.double() if v.is_floating_point() else v) for k, v in student.items()},
                                pseudo_threshold=0.30,
                                teacher_sd={k: (v.double() if v.is_floating_point() else v) for k, v in teacher.items()})
            b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
            random.seed(101 + it); np.random.seed(101 + it)
            _, ex64 = o64.train_step(b64, masks=ex['masks'], return_extras=True,
                                     pseudo_override=(ex['pseudo_label'], ex['n_conf']))
            rows = []
            for name, g64 in ex64['grads'].items():
                e_hip = rel(arena.view(arena.grad, name), g64)
                e_ref = rel(ex['grads'][name], g64)
                rows.append((name, e_hip, e_ref))
            print('grad rel err vs fp64:  HIP / reference-fp32')
            for name, a, b in rows:
                print(f'   {a:.2e} {b:.2e} {name}')
            for name, a, b in rows:
                assert a <= max(TOL, 5.0 * b), (name, a, b)
            for name in ('decode_head.conv_seg.bias', 'auxiliary_head.conv_seg.weight', 'auxiliary_head.conv_seg.bias'):
                assert rel(arena.view(arena.grad, name), ex['grads'][name]) < TOL, name
            flat_64 = torch.cat([g.flatten() for g in ex64['grads'].values()])
            flat_o = torch.cat([g.flatten() for g in ex['grads'].values()])
            flat_m = torch.cat([arena.view(arena.grad, n).flatten() for n in ex['grads']])
            print('flat gradient rel err vs fp64: HIP %.3e  reference-fp32 %.3e' % (rel(flat_m, flat_64), rel(flat_o, flat_64)))
            assert rel(flat_m, flat_64) <= max(TOL, 2.0 * rel(flat_o, flat_64))
    # --- against the reference's own numbers (same seeds as make_golden.py: python/numpy seed 0 BEFORE step 0).
    # Only the first iteration is comparable to 1e-3 (identical weights); see the note on AdamW above.
    model, opt, student, teacher = _build(0.30)
    model.injected_pseudo = None
    random.seed(0); np.random.seed(0)
    batch = synth_batch(2, 128, 6, seed=1234)
    model.debug = {}
    out = model.train_step(to_dev(batch, 'cuda'), opt)
    keys = [str(k) for k in gold['it0_log_keys']]
    vals = gold['it0_log_vals']
    assert list(out['log_vars'].keys()) == keys
    for k, v in zip(keys, vals):
        assert abs(out['log_vars'][k] - v) <= 2 * TOL * max(abs(v), 1e-2), (k, out['log_vars'][k], v)
    # the reference stores vis|seg_mask_mix = where(pseudo_weight > 0, mixed_lbl, 255) (pfgst.py:346-348)
    ml = model.debug['mixed_lbl'].cpu()
    ml = torch.where(model.debug['mixed_w'].cpu().unsqueeze(1) > 0, ml, torch.full_like(ml, 255))
    assert (ml.numpy() != gold['it0_mixed_lbl']).mean() < 2e-3
    arena = model.student_arena
    g = arena.view(arena.grad, 'decode_head.conv_seg.weight')
    assert rel(g, torch.from_numpy(gold['it0_grad|decode_head.conv_seg.weight'])) < 5 * TOL
    g = arena.view(arena.grad, 'backbone.stem.0.weight')
    assert rel(g, torch.from_numpy(gold['it0_grad|backbone.stem.0.weight'])) < 0.1   # fp32 conditioning, see above
    # the reference's own step had a live target side (92 of 512 grid pixels un-mixed with all nine neighbours) ...
    assert float(gold['it0_ignore_mask_trg'].mean()) >= 0.15 and float(vals[keys.index('loss_sim_pos')]) != 0.0
    assert_live_target_side(out['log_vars'])
    # ... and the layers right below the mixed-pass logits carry its gradient (PFGSTLoss -> softmax(logits_trg) -> conv_seg <- CE),
    # against the executed reference: 1e-3 on the classifier (measured 2e-6), 2e-2 one BatchNorm layer down (measured 3e-4 / 5e-3);
    # from two BN layers down the comparison is in the conditioning regime of the stem gradient above (the reference's own fp32
    # gradient is 2.3e-2 from fp64 on the flat vector; this run also uses its OWN pseudo labels, a few near-tie pixels apart) --
    # measured 2-4e-2, bound 0.1.  The per-link test (tests/test_layer_backward_gpu.py) is the tight bound.
    for k in gold.files:
        if k.startswith('it0_grad|') and k[9:] not in ('decode_head.conv_seg.weight', 'backbone.stem.0.weight'):
            gg = arena.view(arena.grad, k[9:])
            e = rel(gg.reshape(gg.shape[0], -1)[:32, :64], torch.from_numpy(gold[k]))
            print(f'   golden gradient sample {k[9:]}: rel err {e:.2e}')
            assert e < (TOL if 'conv_seg' in k else 2e-2 if 'sep_bottleneck.1' in k else 0.1), (k, e)


class count_fused_dgrad_launches:
    """`with count_fused_dgrad_launches() as n:` -- n['f16x3'] / n['bf16x6'] / n['f32'] = data-gradient launches of the step that
    carried the fused BatchNorm-backward epilogue (a `bnb` argument: conv_igemm_f16x3_bnb_kernel and its siblings), n['all'] = every
    data-gradient launch of those entry points"""

    NAMES = dict(f16x3='conv_dgrad_f16x3', bf16x6='conv_dgrad_split', f32='conv_dgrad')

    def __enter__(self):
        from pfst_amd import hip_ops
        self.ops, self.orig, self.n = hip_ops, {}, dict(f16x3=0, bf16x6=0, f32=0, all=0)
        for key, fn in self.NAMES.items():
            self.orig[fn] = getattr(hip_ops, fn)

            def wrapped(*a, _key=key, _fn=self.orig[fn], **kw):
                self.n['all'] += 1
                self.n[_key] += kw.get('bnb') is not None
                return _fn(*a, **kw)
            setattr(hip_ops, fn, wrapped)
        return self.n

    def __exit__(self, *exc):
        for fn, f in self.orig.items():
            setattr(self.ops, fn, f)


def test_train_step_at_512_matches_oracle():
    """One whole step at b = 2 x 512^2 (BASELINE config #1's tile size) under the DEFAULT arithmetic against the oracle.
    At S = 128 a 1/8-grid GEMM has two pixel tiles per image; here it has 32, so the f16x3 machinery of round 3 sits inside an
    end-to-end comparison: workgroups walk tile CHAINS (1024 tiles of layer4.conv3 on 512 resident slots), grids run several rounds,
    the 3x3 kernels use the band tile order per XCD, the whole-line weight gradients split K over pixel chunks, the Winograd-domain
    GEMMs run image-after-image per filter set, and the K >= 512 data gradients carry the fused BatchNorm-backward sums.
    Same checks as the S = 128 test: 14 log values, mixed-label map exact, logits element-wise, gradients against fp64."""
    from oracle import pfst_oracle as O
    from pfst_amd import hip_ops, layers
    from pfst_amd._lib import lib
    from pfst_amd.hostinfo import usable_cpus
    from pfst_amd.synthetic import synth_batch
    assert layers.CONV_MATH == os.environ.get('PFST_CONV_MATH', 'f16x3')
    b, S = 2, 512
    # the machinery engages at this size: more tiles than workgroups for the large 1x1 GEMMs -> chains
    tiles = (S // 8) * (S // 8) // 128 * (2048 // 128) * b            # layer4.conv3 in 128 x 128 tiles (the unit pfst_f16x3_chain_grid counts slots in)
    if layers.CONV_MATH == 'f16x3':
        assert lib().pfst_f16x3_chain_grid(tiles, 1) < tiles, 'layer4.conv3 would run one tile per workgroup: no chain in this test'
    model, opt, student, teacher = _build(0.30)
    oracle = O.OraclePFGST(student, pseudo_threshold=0.30, teacher_sd=teacher)
    torch.set_num_threads(usable_cpus())
    batch = synth_batch(b, S, 6, seed=4242)
    random.seed(101); np.random.seed(101)
    olog, ex = oracle.train_step(batch, return_extras=True)
    random.seed(101); np.random.seed(101)
    model.debug = {}
    model.injected_pseudo = (ex['pseudo_label'].to(torch.uint8).cuda(), torch.tensor([ex['n_conf']], dtype=torch.int64).cuda())
    with count_fused_dgrad_launches() as launches:
        out = model.train_step(to_dev(batch, 'cuda'), opt)
    dbg, lv = model.debug, out['log_vars']
    print(f'data-gradient launches {launches}')
    if layers.FUSE_BN_BWD and layers.CONV_MATH == 'f16x3' and layers.FUSE_BN_BWD_MIN_K_SPLIT <= 512:
        assert launches['f16x3'] >= 30, launches          # per student graph: conv3 of layer2-4, conv1 of layer4, ASPP pointwise + 1x1, sep_bottleneck.1
    assert list(lv.keys()) == list(olog.keys()) and out['num_samples'] == b
    mism = 1.0 - (dbg['own_pseudo_label'].cpu() == ex['pseudo_label']).float().mean().item()
    print(f'end-to-end pseudo-label mismatch rate {mism:.2e}')
    assert mism < 2e-3, mism
    l64, _, _ = hip_ops.pseudo_label(ex['ema_logits_low'].cuda(), (S, S), 0.30)
    assert torch.equal(l64.cpu(), ex['pseudo_label']), 'pseudo-label kernel must be bit exact on identical logits'
    assert torch.equal(dbg['mix_masks'].cpu().long(), ex['masks']), 'class-mix masks'
    assert bool((dbg['mixed_lbl'].cpu() == ex['mixed_lbl']).all()), 'mixed label map'
    assert abs(int(dbg['own_conf_count'].item()) - ex['n_conf']) <= 16
    for name, key in (('source logits', 'src_logits'), ('mixed-pass logits', 'mix_logits'), ('teacher decoded features', 'ema_dec')):
        assert rel(dbg[key], ex[key]) < TOL, (name, rel(dbg[key], ex[key]))
        assert_elementwise(dbg[key], ex[key], name)
    assert_elementwise(dbg['ema_logits'], ex['ema_logits_low'], 'teacher logits')
    assert rel(dbg['mixed_w'], ex['mixed_w']) < 1e-5
    for k in olog:
        assert abs(lv[k] - olog[k]) <= TOL * max(abs(olog[k]), 1e-2), (k, lv[k], olog[k])
    assert_live_target_side(olog, ex)
    assert_live_target_side(lv)
    # gradients: as close to fp64 as the oracle's own fp32 path (x5), 1e-3 next to the loss -- the criterion of the S = 128 test
    arena = model.student_arena
    dd = lambda sd: {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    o64 = O.OraclePFGST(dd(student), pseudo_threshold=0.30, teacher_sd=dd(teacher))
    b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    random.seed(101); np.random.seed(101)
    _, ex64 = o64.train_step(b64, masks=ex['masks'], return_extras=True, pseudo_override=(ex['pseudo_label'], ex['n_conf']))
    rows = [(name, rel(arena.view(arena.grad, name), g64), rel(ex['grads'][name], g64)) for name, g64 in ex64['grads'].items()]
    print('grad rel err vs fp64:  HIP / oracle-fp32 (worst ten by ratio)')
    for name, a, r in sorted(rows, key=lambda t: -t[1] / max(t[2], 1e-12))[:10]:
        print(f'   {a:.2e} {r:.2e} {name}')
    for name, a, r in rows:
        assert a <= max(TOL, 5.0 * r), (name, a, r)
    for name in ('decode_head.conv_seg.bias', 'auxiliary_head.conv_seg.weight', 'auxiliary_head.conv_seg.bias'):
        assert rel(arena.view(arena.grad, name), ex['grads'][name]) < TOL, name
    flat_64 = torch.cat([g.flatten() for g in ex64['grads'].values()])
    flat_o = torch.cat([g.flatten() for g in ex['grads'].values()])
    flat_m = torch.cat([arena.view(arena.grad, n).flatten() for n in ex['grads']])
    print('flat gradient rel err vs fp64: HIP %.3e  oracle-fp32 %.3e' % (rel(flat_m, flat_64), rel(flat_o, flat_64)))
    assert rel(flat_m, flat_64) <= max(TOL, 2.0 * rel(flat_o, flat_64))
    # ... and against the EXECUTED reference at this size (tests/golden/train_step_512.npz, make_golden.py:gen_train_step_512, same seeds;
    # tests/test_oracle_golden.py::test_train_step_512_golden pins the oracle to the same fixture): log values, the mixed label map as the
    # reference stores it (255 where the pixel weight is zero), gradients next to the loss
    gold = np.load(os.path.join(G, 'train_step_512.npz'))
    assert list(lv.keys()) == [str(k) for k in gold['log_keys']]
    for k, v in zip(lv, gold['log_vals']):
        assert abs(lv[k] - v) <= 2 * TOL * max(abs(v), 1e-2), (k, lv[k], v)
    ml = dbg['mixed_lbl'].cpu()
    ml = torch.where(dbg['mixed_w'].cpu().unsqueeze(1) > 0, ml, torch.full_like(ml, 255))
    assert (ml.numpy().astype(np.uint8) != gold['mixed_lbl']).mean() < 1e-9          # the step ran on the oracle's (= the reference's) pseudo labels
    for name in ('decode_head.conv_seg.bias', 'decode_head.conv_seg.weight', 'auxiliary_head.conv_seg.weight'):
        gg = arena.view(arena.grad, name)
        e = rel(gg.reshape(gg.shape[0], -1)[:32, :64], torch.from_numpy(gold['grad|' + name]))
        print(f'   golden gradient sample {name}: rel err {e:.2e}')
        assert e < 5 * TOL, (name, e)


def test_pfgst_loss_downscale1_matches_oracle():
    """SeasonNet setting (configs/pfst/pfst_season_net_sp2fa_*.py: downscale=1): 1/8 features resized to the 1/4 grid."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd.engine import Tape, Var
    from pfst_amd.registry import build_loss
    from pfst_amd import hip_ops
    g = torch.Generator().manual_seed(3)
    B, C, S = 2, 33, 128
    lt = (torch.randn(B, C, S // 4, S // 4, generator=g) * 2).requires_grad_()
    xe = torch.randn(B, 24, S // 8, S // 8, generator=g)
    xs = torch.randn(B, 24, S // 8, S // 8, generator=g).requires_grad_()
    gts = torch.randint(0, C, (B, 1, 4, 4), generator=g).repeat_interleave(S // 4, 2).repeat_interleave(S // 4, 3)
    gts[:, :, :8, :8] = 255
    mm = (torch.rand(B, 1, 2, 2, generator=g) > 0.6).long().repeat_interleave(S // 2, 2).repeat_interleave(S // 2, 3)
    ref, _ = O.pfgst_loss(lt, xe, xs, gts, mm, O.DEFAULT_LOSS_W, downscale=1)
    sum(v.sum() for v in ref.values()).backward()
    L = build_loss(dict(type='PFGSTLoss', kernel_size=3, dilation=2, top_k=3, weights=O.DEFAULT_LOSS_W, sim_type='cosine',
                        feat_level=None, detach_unfold=True, downscale=1))
    tape = Tape()
    ltv, xsv = Var(lt.detach().cuda(), True), Var(xs.detach().cuda(), True)
    out = L(dict(logits_trg=ltv, x_ema=Var(xe.cuda()), x_src=xsv, gt_src=hip_ops.to_u8(gts.cuda()),
                 mix_masks=hip_ops.to_u8(mm.cuda())), tape=tape)
    tape.backward()
    for k, v in ref.items():
        assert abs(float(out[k]) - float(v.sum())) <= 1e-4 * max(abs(float(v.sum())), 1e-3), k
    assert rel(xsv.grad, xs.grad) < TOL
    assert rel(ltv.grad, lt.grad) < TOL


def test_seasonnet_like_step_10band_33class():
    """BASELINE config #5 shape (C=33, 10 input bands, downscale=1) at reduced size: one step vs the oracle."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd.optim import build_optimizer
    from pfst_amd.presets import uda_cfg as preset_cfg
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import synth_batch
    cfg = preset_cfg(33, 10, dropout=0.0, blur=False, color_jitter_probability=2.0, downscale=1, pseudo_threshold=0.05)
    model = UDA.build(cfg)
    both, student, teacher = seeded_pfgst_state(O, 4, 33, 10)
    model.load_state_dict(both, strict=False)
    model.cuda()
    opt = build_optimizer(model, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01))
    batch = synth_batch(2, 128, 33, cin=10, seed=77)
    oracle = O.OraclePFGST(student, pseudo_threshold=0.05, teacher_sd=teacher, downscale=1)
    random.seed(5); np.random.seed(5)
    olog, ex = oracle.train_step(batch, return_extras=True)
    random.seed(5); np.random.seed(5)
    model.debug = {}
    model.injected_pseudo = (ex['pseudo_label'].to(torch.uint8).cuda(), torch.tensor([ex['n_conf']], dtype=torch.int64).cuda())
    out = model.train_step(to_dev(batch, 'cuda'), opt)
    for k, v in olog.items():
        # acc_seg counts arg-max hits: a handful of near-tie pixels (of 32768) may flip -> absolute slack of 10 pixels
        tol = 100.0 * 10 / (2 * 128 * 128) if k.endswith('acc_seg') else TOL * max(abs(v), 1e-2)
        assert abs(out['log_vars'][k] - v) <= tol, (k, out['log_vars'][k], v)
    assert rel(model.debug['mix_logits'], ex['mix_logits']) < TOL
    assert (model.debug['own_pseudo_label'].cpu() != ex['pseudo_label']).float().mean() < 5e-3
    assert_live_target_side(olog, ex)


def test_inria_like_binary_step_with_part_threshold():
    """BASELINE config #4 shape (C=2, binary building segmentation) at reduced size + thre_type='part' (SURVEY §8 f4):
    per-pixel confidence weights instead of the scalar fraction."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd.optim import build_optimizer
    from pfst_amd.presets import uda_cfg as preset_cfg
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import synth_batch
    cfg = preset_cfg(2, 3, dropout=0.0, blur=False, color_jitter_probability=2.0, pseudo_threshold=0.52)
    cfg['thre_type'] = 'part'
    model = UDA.build(cfg)
    both, student, teacher = seeded_pfgst_state(O, 6, 2, 3)
    model.load_state_dict(both, strict=False)
    model.cuda()
    opt = build_optimizer(model, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01))
    batch = synth_batch(2, 128, 2, seed=99)
    oracle = O.OraclePFGST(student, pseudo_threshold=0.52, teacher_sd=teacher, thre_type='part')
    random.seed(8); np.random.seed(8)
    olog, ex = oracle.train_step(batch, return_extras=True)
    random.seed(8); np.random.seed(8)
    model.debug = {}
    out = model.train_step(to_dev(batch, 'cuda'), opt)
    dbg = model.debug
    same = dbg['pseudo_label'].cpu() == ex['pseudo_label']
    assert 1 - same.float().mean() < 5e-3
    # per-pixel weights: identical wherever the confidence decision is not within rounding of the threshold
    wdiff = (dbg['mixed_w'].cpu() != ex['mixed_w']).float().mean()
    assert wdiff < 5e-3, wdiff
    assert 0.02 < float(ex['mixed_w'].mean()) < 0.999          # the mask is non-trivial
    for k, v in olog.items():
        tol = 100.0 * 40 / (2 * 128 * 128) if k.endswith('acc_seg') else 5e-3 * max(abs(v), 1e-2)
        assert abs(out['log_vars'][k] - v) <= tol, (k, out['log_vars'][k], v)
    assert_live_target_side(olog, ex)


def test_step_with_pfgst_loss_option_variants():
    """One full train step with the non-default PFGSTLoss options (gaussian similarity, squared-hinge source
    losses, gradient through the unfolded probabilities, all nine pairs) against the oracle."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd.optim import build_optimizer
    from pfst_amd.presets import uda_cfg as preset_cfg
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import synth_batch
    opts = dict(sim_type='gaussian', sigma=20.0, top_k=None, detach_unfold=False, src_loss_type='margin2', margin=(0.7, 0.2), src_perc=0.7)
    cfg = preset_cfg(6, 3, dropout=0.0, blur=False, color_jitter_probability=2.0, pseudo_threshold=0.3)
    cfg['aux_losses'][0].update(opts)
    model = UDA.build(cfg)
    both, student, teacher = seeded_pfgst_state(O, 9)
    model.load_state_dict(both, strict=False)
    model.cuda()
    opt = build_optimizer(model, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01))
    batch = synth_batch(2, 128, 6, seed=77)
    oracle = O.OraclePFGST(student, pseudo_threshold=0.3, teacher_sd=teacher, loss_opts=opts)
    random.seed(3); np.random.seed(3)
    olog, ex = oracle.train_step(batch, return_extras=True)
    random.seed(3); np.random.seed(3)
    out = model.train_step(to_dev(batch, 'cuda'), opt)
    assert set(olog) == set(out['log_vars']) and 'loss_src_pos' in olog and 'loss_src_pos_mean' not in olog
    for k, v in olog.items():
        tol = 100.0 * 40 / (2 * 128 * 128) if k.endswith('acc_seg') else 5e-3 * max(abs(v), 1e-2)
        assert abs(out['log_vars'][k] - v) <= tol, (k, out['log_vars'][k], v)
    assert_live_target_side(olog, ex)


@pytest.mark.parametrize('level', [3, 1])
def test_step_with_backbone_feature_level(level):
    """use_decoded_feats=False + PFGSTLoss(feat_level=k) (pfgst.py:229-231,255-257; pfgst_loss.py:50-51): the similarity losses act
    on backbone feature map k (2048 / 512 channels at 1/8) instead of the decoded features, and their gradient enters the
    backbone there.  Losses and the gradient of the first backbone tensor upstream of that map against the oracle."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd.optim import build_optimizer
    from pfst_amd.presets import uda_cfg as preset_cfg
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import synth_batch
    cfg = preset_cfg(6, 3, dropout=0.0, blur=False, color_jitter_probability=2.0, pseudo_threshold=0.3)
    cfg['use_decoded_feats'] = False
    cfg['aux_losses'][0]['feat_level'] = level
    # 1000x the shipped loss weights: the similarity losses then dominate the gradient where they enter the backbone, one
    # well-conditioned BN layer above the tensor checked below (through the heads' ~10 BN layers fp32 noise alone is 2-5 %)
    aux_w = {k: 100.0 for k in cfg['aux_losses'][0]['weights']}
    cfg['aux_losses'][0]['weights'] = dict(aux_w)
    model = UDA.build(cfg)
    both, student, teacher = seeded_pfgst_state(O, 9)
    model.load_state_dict(both, strict=False)
    model.cuda()
    opt = build_optimizer(model, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01))
    batch = synth_batch(2, 128, 6, seed=78)
    oracle = O.OraclePFGST(student, pseudo_threshold=0.3, teacher_sd=teacher, feat_level=level, aux_weights=aux_w)
    random.seed(3); np.random.seed(3)
    olog, ex = oracle.train_step(batch, return_extras=True)
    o_dec = O.OraclePFGST(student, pseudo_threshold=0.3, teacher_sd=teacher, aux_weights=aux_w)    # decoded-feature losses differ
    random.seed(3); np.random.seed(3)
    dlog, dex = o_dec.train_step(batch, return_extras=True)
    assert abs(dlog['loss_src_pos_mean'] - olog['loss_src_pos_mean']) > 1e-3 * abs(olog['loss_src_pos_mean'])
    random.seed(3); np.random.seed(3)
    out = model.train_step(to_dev(batch, 'cuda'), opt)
    assert set(olog) == set(out['log_vars'])
    for k, v in olog.items():
        tol = 100.0 * 40 / (2 * 128 * 128) if k.endswith('acc_seg') else 5e-3 * max(abs(v), 1e-2)
        assert abs(out['log_vars'][k] - v) <= tol, (k, out['log_vars'][k], v)
    assert_live_target_side(olog, ex)
    # backward: the loss gradient enters the backbone at feature map `level`; the tensor producing that map must carry it --
    # the HIP gradient is far closer to the feature-level oracle than the decoded-feature oracle's gradient is
    name = {3: 'backbone.layer4.2.bn3.weight', 1: 'backbone.layer2.3.bn3.weight'}[level]
    a = model.student_arena
    e_hip = rel(a.view(a.grad, name), ex['grads'][name])
    e_other = rel(dex['grads'][name], ex['grads'][name])
    print(f'feat_level={level}: grad rel err vs oracle {e_hip:.2e}; decoded-feature oracle differs by {e_other:.2e}')
    assert e_hip < 6e-2 and e_other > 0.5, (name, e_hip, e_other)      # measured 1.0e-2 (level 3), 2.4e-2 (level 1) vs 8.3 / 36.8
    # a tuple of features with feat_level=None is a type error in the reference too (F.unfold of a tuple)
    cfg['aux_losses'][0]['feat_level'] = None
    bad = UDA.build(cfg)
    bad.load_state_dict(both, strict=False)
    bad.cuda()
    with pytest.raises(TypeError):
        bad.train_step(to_dev(batch, 'cuda'), build_optimizer(bad, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)))


def test_stream_overlap_options_do_not_change_the_step():
    """PFST_WGRAD_STREAM / PFST_FORK_TEACHER (layers.set_overlap; the product default since round 5): weight gradients on a side stream and
    the teacher's forward forked beside the student's source pass are pure scheduling -- the step gives the same pseudo labels, losses and gradients
    as the single-stream schedule (to the fp32 atomics' summation-order noise); a second step (after AdamW, which amplifies
    that noise, see above) stays close."""
    from pfst_amd import layers
    from pfst_amd.synthetic import synth_batch
    batch = to_dev(synth_batch(2, 128, 6, seed=77), 'cuda')
    runs = []
    default = (layers.WGRAD_STREAM, layers.FORK_TEACHER)
    for overlap in (False, True):
        model, opt, _, _ = _build(0.30)
        model.debug = {}
        layers.set_overlap(overlap, overlap)
        try:
            random.seed(106); np.random.seed(106)      # a class draw that leaves a live target region (helpers.assert_live_target_side)
            log0 = model.train_step(batch, opt)['log_vars']
            grad0 = model.student_arena.grad.clone().cpu()
            pl0 = model.debug['pseudo_label'].cpu()
            random.seed(101); np.random.seed(101)
            log1 = model.train_step(batch, opt)['log_vars']
            torch.cuda.synchronize()
        finally:
            layers.set_overlap(*default)
        runs.append((log0, pl0, grad0, log1, model._teacher_arena.data.clone().cpu()))
    assert layers._side_stream is not None and layers._teacher_stream is not None, 'the overlap streams were never used'
    (a0, p0, g0, a1, t0), (b0, p1, g1, b1, t1) = runs
    assert torch.equal(p0, p1), 'pseudo labels'
    assert list(a0) == list(b0)
    assert_live_target_side(a0)
    for k in a0:
        assert abs(a0[k] - b0[k]) <= 1e-5 * max(abs(a0[k]), 1e-2), (k, a0[k], b0[k])
        assert abs(a1[k] - b1[k]) <= 5e-3 * max(abs(a1[k]), 1e-1), (k, a1[k], b1[k])
    assert rel(g1, g0) < 1e-4, rel(g1, g0)
    assert rel(t1, t0) < 1e-5


def test_side_stream_reads_survive_freed_host_references():
    """Every tensor a queued side-stream launch reads must be recorded on that stream: the host drops its references when a backward closure
    returns, long before the weight-gradient launch it queued runs, and the caching allocator hands the block to the next allocation of the
    main stream.  Round 5: the coefficient rows of a normalise-on-load input (layers.FOLD_BN_GEMM / FOLD_BN_DWSEP) were not -- the third model
    built in one process (recycled, no longer zero-filled memory) produced NaN weight gradients for 485 input channels of an ASPP pointwise
    layer.  Four models built one after another, same seeds: every step must reproduce the first model's gradients."""
    from pfst_amd.synthetic import synth_batch
    batch = to_dev(synth_batch(2, 128, 6, seed=1234), 'cuda')
    grads = []
    for _ in range(4):
        model, opt, _, _ = _build(0.30)
        random.seed(106); np.random.seed(106)
        model.train_step(batch, opt)
        g = model.student_arena.grad.clone().cpu()
        assert bool(torch.isfinite(g).all()), int((~torch.isfinite(g)).sum())
        grads.append(g)
        del model, opt
    for g in grads[1:]:
        assert rel(g, grads[0]) < 1e-4, rel(g, grads[0])


def test_fused_bn_backward_does_not_change_the_step():
    """layers.FUSE_BN_BWD: the BatchNorm-backward sums come out of the epilogue of the data-gradient launch that completes dL/dy
    (csrc/conv_epilogue.h) instead of a reduction pass -- same mathematics, so one whole train step (both student graphs, as
    wired: accumulate epilogues, residual gates, the final-writer logic of engine.Var) must give the same gradients and the same
    second-step losses with the fusion on and off."""
    from pfst_amd import layers
    from pfst_amd.synthetic import synth_batch
    batch = to_dev(synth_batch(2, 128, 6, seed=77), 'cuda')
    runs = []
    prev = layers.FUSE_BN_BWD
    for fuse in (False, True):
        layers.FUSE_BN_BWD = fuse
        try:
            model, opt, _, _ = _build(0.30)
            random.seed(106); np.random.seed(106)      # a class draw that leaves a live target region (helpers.assert_live_target_side)
            with count_fused_dgrad_launches() as launches:
                log0 = model.train_step(batch, opt)['log_vars']
            grad0 = model.student_arena.grad.clone().cpu()
            random.seed(101); np.random.seed(101)
            log1 = model.train_step(batch, opt)['log_vars']
        finally:
            layers.FUSE_BN_BWD = prev
        runs.append((log0, grad0, log1))
        # the comparison is only worth something if the fused epilogue really ran in the `fuse` run, under the arithmetic in use
        # (default f16x3: conv_igemm_f16x3_bnb_kernel), and not at all in the other
        print(f'FUSE_BN_BWD={fuse}: data-gradient launches {launches}')
        if fuse:
            # per student graph: conv3 of layer2-4 (13), conv1 of layer4 (3), the three ASPP pointwise convs + the 1x1 branch, sep_bottleneck.1
            assert launches[layers.CONV_MATH] >= 30, launches
        else:
            assert launches['f16x3'] + launches['bf16x6'] + launches['f32'] == 0, launches
    (a0, g0, a1), (b0, g1, b1) = runs
    assert_live_target_side(a0)           # the whole loss graph is alive: the target-side PFGSTLoss gradient runs through the fused sums too
    for k in a0:
        assert abs(a0[k] - b0[k]) <= 1e-6 * max(abs(a0[k]), 1e-2), (k, a0[k], b0[k])       # forward is untouched
        assert abs(a1[k] - b1[k]) <= 5e-3 * max(abs(a1[k]), 1e-1), (k, a1[k], b1[k])
    assert rel(g1, g0) < 2e-4, rel(g1, g0)     # fp32 partial sums in a different order, amplified by the BN chain (atomics noise alone: 1e-4)


def test_vis_states_follow_the_reference_layout():
    """pfgst.py:335,346-352 + pfgst_loss.py:134-137: `states` holds three tuples of live tensors.  Off by default (they cost
    full-resolution passes nothing on the path needs); with `return_vis_states` the reference's layout and values."""
    from pfst_amd.synthetic import synth_batch
    gold = np.load(os.path.join(G, 'train_step.npz'))
    model, opt, student, teacher = _build(0.30)
    batch = to_dev(synth_batch(2, 128, 6, seed=1234), 'cuda')
    random.seed(0); np.random.seed(0)
    assert model.train_step(batch, opt)['states'] == {}
    model, opt, student, teacher = _build(0.30)
    model.return_vis_states = True
    random.seed(0); np.random.seed(0)
    st = model.train_step(batch, opt)['states']
    assert set(st) == {'vis|density_sim_feat', 'vis|seg_mask_src', 'vis|seg_mask_mix'}
    shapes = {k: [tuple(t.shape) for t in v] for k, v in st.items()}
    assert shapes['vis|density_sim_feat'] == [(2, 3, 128, 128), (2, 1, 16, 16), (2, 1, 16, 16)]
    assert shapes['vis|seg_mask_src'] == [(2, 3, 128, 128), (2, 1, 128, 128), (2, 1, 32, 32)]
    assert shapes['vis|seg_mask_mix'] == [(2, 3, 128, 128), (2, 1, 128, 128), (2, 1, 32, 32)]
    assert st['vis|density_sim_feat'][2].dtype == torch.bool and st['vis|seg_mask_mix'][2].dtype == torch.float32
    assert (st['vis|seg_mask_mix'][1].cpu().numpy() != gold['it0_mixed_lbl']).mean() < 2e-3
    assert (st['vis|seg_mask_mix'][2].cpu().numpy().astype(np.int64) != gold['it0_mix_pred']).mean() < 5e-3


def test_pseudo_weight_ignore_rows_and_invalid_labels():
    """pseudo_weight_ignore_top / _bottom (pfgst.py:273-276) against the oracle; a label outside [0, C) that is not ignore_index
    raises after the step's single read, as F.cross_entropy does in the reference."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd.optim import build_optimizer
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import synth_batch
    cfg = uda_cfg(threshold=0.30)
    cfg['pseudo_weight_ignore_top'], cfg['pseudo_weight_ignore_bottom'] = 16, 8
    model = UDA.build(cfg)
    both, student, teacher = seeded_pfgst_state(O, 9)
    model.load_state_dict(both, strict=False)
    model.cuda()
    opt = build_optimizer(model, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01))
    batch = synth_batch(2, 128, 6, seed=55)
    oracle = O.OraclePFGST(student, pseudo_threshold=0.30, teacher_sd=teacher, ignore_top=16, ignore_bottom=8)
    random.seed(4); np.random.seed(4)
    olog, ex = oracle.train_step(batch, return_extras=True)
    random.seed(4); np.random.seed(4)
    model.debug = {}
    model.injected_pseudo = (ex['pseudo_label'].to(torch.uint8).cuda(), torch.tensor([ex['n_conf']], dtype=torch.int64).cuda())
    out = model.train_step(to_dev(batch, 'cuda'), opt)
    assert torch.equal(model.debug['mixed_w'].cpu(), ex['mixed_w'])
    assert float(ex['mixed_w'][:, :16][ex['masks'][:, 0, :16] == 0].abs().max()) == 0.0          # the rows really are zeroed
    for k, v in olog.items():
        assert abs(out['log_vars'][k] - v) <= TOL * max(abs(v), 1e-2), (k, out['log_vars'][k], v)
    assert_live_target_side(olog, ex)
    bad = to_dev(synth_batch(2, 128, 6, seed=56), 'cuda')
    bad['gt_semantic_seg'][0, 0, 40:44, 40:44] = 7
    with pytest.raises(ValueError, match='outside'):
        model.train_step(bad, opt)


def test_step_with_trainable_projection():
    """PFGSTLoss(proj_net_cfg=...) inside a whole train step (pfgst_loss.py:34-36,73-75): the projection is a parameter of the UDA
    module (state_dict key aux_losses.0.proj_net.*), receives gradient from the source AND the teacher branch, and is stepped by
    AdamW with the student.  Losses and the projection's gradient against the oracle."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd.optim import build_optimizer
    from pfst_amd.presets import uda_cfg as preset_cfg
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import synth_batch
    cfg = preset_cfg(6, 3, dropout=0.0, blur=False, color_jitter_probability=2.0, pseudo_threshold=0.3)
    cfg['aux_losses'][0]['proj_net_cfg'] = dict(in_channels=512, out_channels=64)
    model = UDA.build(cfg)
    both, student, teacher = seeded_pfgst_state(O, 9)
    model.load_state_dict(both, strict=False)
    assert 'aux_losses.0.proj_net.weight' in model.state_dict() and 'aux_losses.0.proj_net.bias' in model.state_dict()
    model.cuda()
    pw, pb = model.aux_losses[0].proj_net.weight.detach().cpu().clone(), model.aux_losses[0].proj_net.bias.detach().cpu().clone()
    opt = build_optimizer(model, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01))
    batch = synth_batch(2, 128, 6, seed=77)
    oracle = O.OraclePFGST(student, pseudo_threshold=0.3, teacher_sd=teacher, proj=(pw, pb))
    random.seed(3); np.random.seed(3)
    olog, ex = oracle.train_step(batch, return_extras=True)
    random.seed(3); np.random.seed(3)
    model.injected_pseudo = (ex['pseudo_label'].to(torch.uint8).cuda(), torch.tensor([ex['n_conf']], dtype=torch.int64).cuda())
    out = model.train_step(to_dev(batch, 'cuda'), opt)
    for k, v in olog.items():
        tol = 100.0 * 40 / (2 * 128 * 128) if k.endswith('acc_seg') else 5e-3 * max(abs(v), 1e-2)
        assert abs(out['log_vars'][k] - v) <= tol, (k, out['log_vars'][k], v)
    assert_live_target_side(olog, ex)
    proj = model.aux_losses[0].proj_net
    assert rel(proj.weight.grad, ex['proj_grads'][0]) < 2e-2, rel(proj.weight.grad, ex['proj_grads'][0])     # one BN layer above the features
    assert rel(proj.bias.grad, ex['proj_grads'][1]) < 2e-2
    assert not torch.equal(proj.weight.detach().cpu(), pw)                     # AdamW stepped it
    assert rel(proj.weight, oracle.proj[0]) < 1e-3


def test_step_with_apply_no_mix():
    """apply_no_mix (pfgst.py:283-289): the mix classes are still drawn (same NumPy stream), the masks are zeroed and the UN-augmented target
    image is used, so the 'mixed' pass is a pure target pass under pseudo labels; one full step against the oracle."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd.optim import build_optimizer
    from pfst_amd.presets import uda_cfg as preset_cfg
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import synth_batch
    cfg = preset_cfg(6, 3, dropout=0.0, blur=False, color_jitter_probability=2.0, pseudo_threshold=0.3)
    cfg['apply_no_mix'] = True
    model = UDA.build(cfg)
    both, student, teacher = seeded_pfgst_state(O, 12)
    model.load_state_dict(both, strict=False)
    model.cuda()
    opt = build_optimizer(model, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01))
    batch = synth_batch(2, 128, 6, seed=41)
    oracle = O.OraclePFGST(student, pseudo_threshold=0.3, teacher_sd=teacher, apply_no_mix=True)
    random.seed(5); np.random.seed(5)
    olog, ex = oracle.train_step(batch, return_extras=True)
    state_after = np.random.get_state()[1][:4].copy()
    random.seed(5); np.random.seed(5)
    model.debug = {}
    out = model.train_step(to_dev(batch, 'cuda'), opt)
    assert np.array_equal(np.random.get_state()[1][:4], state_after)          # the classes were drawn all the same
    dbg = model.debug
    assert int(dbg['mix_masks'].abs().sum()) == 0 and int(ex['masks'].abs().sum()) == 0
    assert torch.equal(dbg['mixed_img'].cpu(), batch['target_img'])           # not the strongly augmented copy
    assert torch.equal(ex['mixed_img'], batch['target_img'])
    same = dbg['mixed_lbl'].cpu() == ex['mixed_lbl']
    assert 1 - same.float().mean() < 5e-3
    for k, v in olog.items():
        tol = 100.0 * 40 / (2 * 128 * 128) if k.endswith('acc_seg') else 5e-3 * max(abs(v), 1e-2)
        assert abs(out['log_vars'][k] - v) <= tol, (k, out['log_vars'][k], v)
    assert_live_target_side(olog, ex)


def test_batch_of_one_fails_like_the_reference():
    """The ASPP image-pool BatchNorm normalises over the batch only (N x 512 x 1 x 1): with one sample per GPU torch's batch_norm raises
    'Expected more than 1 value per channel when training'; so does this path, before any statistics are formed."""
    import pytest
    import pfst_amd  # noqa: F401
    from pfst_amd.optim import build_optimizer
    from pfst_amd.presets import uda_cfg as preset_cfg
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import synth_batch
    model = UDA.build(preset_cfg(6, 3, dropout=0.0, blur=False, color_jitter_probability=2.0))
    model.cuda()
    opt = build_optimizer(model, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01))
    with pytest.raises(ValueError, match='Expected more than 1 value per channel when training'):
        model.train_step(to_dev(synth_batch(1, 64, 6, seed=3), 'cuda'), opt)
