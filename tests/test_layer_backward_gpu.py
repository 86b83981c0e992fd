"""Backward parity of the network AS WIRED, one link at a time (VERDICT r1 'What's weak' #1).

End to end, the backward pass through ~70 train-mode BatchNorm layers of a random-init network amplifies fp32 rounding noise
to 2-5 % per tensor (for the reference's own fp32 path too, tests/test_train_step_gpu.py), so an end-to-end comparison
cannot bound the kernels tighter than that.  This test bounds every link instead:

  * the CPU oracle runs one full EncoderDecoder.forward_train + backward with every named activation captured
    (oracle.CAPTURE), which gives each layer's realistic upstream gradient dL/dy;
  * the HIP model runs its normal forward (slices of the concat buffers, accumulate epilogues, ReLU bitmask gates,
    Winograd saved-V, fused BN statistics -- nothing is re-wired for the test) and its normal tape backward; a
    Tape observer overwrites, right before each closure runs, that closure's incoming gradient buffer IN PLACE with the
    oracle's dL/dy, and measures what the closure adds to its input-gradient buffers and to the parameter-gradient arena;
  * the expected contribution is the oracle's restatement of that one layer (conv -> train-mode BN -> [+ residual] -> ReLU,
    F.max_pool2d, F.interpolate, ...) differentiated by torch autograd on the CPU in fp64, evaluated at the layer's actual HIP
    input and with the HIP forward's ReLU gate.

Bound enforced per link and tensor, ELEMENT-WISE: |hip - ref| <= 1e-4 * max|ref| + 1e-3 * |ref|   (north_star: 1e-3 rel fp32).
"""
import pytest
import torch
import torch.nn.functional as F

from helpers import model_cfg, seeded_pfgst_state

pytestmark = pytest.mark.gpu

RTOL, ATOL_REL = 1e-3, 1e-4


def mixed_err(a, ref):
    """worst |a-ref| / (ATOL_REL*max|ref| + RTOL*|ref|) over the elements (<= 1 passes), and the norm-wise relative error"""
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    scale = float(ref.abs().max())
    if scale == 0.0:
        return float((a.abs().max() > 0)), 0.0
    bound = ATOL_REL * scale + RTOL * ref.abs()
    return float(((a - ref).abs() / bound).max()), float((a - ref).norm() / (ref.norm() + 1e-300))


def _grad_of(v):
    g = v.grad
    return None if g is None else g.clone()


def _data(v):
    """the tensor a Var stands for; a deferred conv -> BN -> ReLU output (never written by the product: its consumer normalises on load) is
    materialised here by the normalisation pass itself -- the same fma / max per element the consumers apply"""
    if v.lazy is None:
        return v.data
    from pfst_amd import hip_ops as ops
    pre, coef, bn = v.lazy
    if bn is None:          # a concat buffer with a coefficient table (the ASPP head's, layers.FOLD_BN_CONCAT): rows (mean, invstd, sc, sh) per channel
        # (the kernels' fma rounds once: the product and sum in fp64, then one rounding to fp32)
        return torch.relu((pre.double() * coef[:, 2].double().view(1, -1, 1, 1) + coef[:, 3].double().view(1, -1, 1, 1)).float())
    return ops.bn_apply(pre, coef[:, 0].contiguous(), coef[:, 1].contiguous(), bn.weight.data, bn.bias.data, not v.lazy_norelu)


@pytest.mark.parametrize('math', ['f32', 'bf16x6', 'f16x3'])
@pytest.mark.parametrize('wino,fold', [(True, False), (False, False), (True, True)], ids=['wino', 'direct', 'wino-deferred'])
def test_every_backward_link_as_wired(wino, fold, math):
    _every_backward_link_as_wired(wino, fold, math)


def test_every_backward_link_as_wired_at_256():
    """the same check on 256^2 tiles (4 x the pixels per plane: 64 x 64 ... 32 x 32 feature maps, whole Winograd tile rows, longer BatchNorm
    reductions) for the product wiring: default arithmetic, Winograd layers, every deferred normalisation on (VERDICT r4 weak #1a)"""
    _every_backward_link_as_wired(True, True, 'f16x3', S=256)


def test_every_backward_link_as_wired_at_512():
    """... and on 512^2 tiles (64 x 64 planes at 1/8 resolution: the strip depthwise kernels, split-K weight gradients over several chunks, the
    stem's 1024-thread statistics rows), same wiring"""
    _every_backward_link_as_wired(True, True, 'f16x3', S=512)


def _every_backward_link_as_wired(wino, fold, math, S=128):
    """fold: the deferred normalisations of the product ON -- stem.6 -> max-pool, sep_bottleneck[0] -> [1] and (round 5) bn1 of the twelve
    Winograd bottlenecks normalised by conv2's input transform: the links whose input or output is never written are checked at the tensor
    the normalisation pass would have written (_data)"""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd import hip_ops as ops
    from pfst_amd import layers
    from pfst_amd.engine import ParamArena, Tape
    from pfst_amd.registry import build_segmentor
    from pfst_amd.synthetic import synth_batch

    C, b = 6, 2
    _, student, _ = seeded_pfgst_state(O, 9)
    batch = synth_batch(b, S, C, seed=4321)
    img, gt = batch['img'], batch['gt_semantic_seg']
    pixw = 0.25 + 0.75 * torch.rand(b, S, S, generator=torch.Generator().manual_seed(5))     # mixed-pass style pixel weights

    # ---- oracle: full forward + backward, upstream gradient of every named activation
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone()) for k, v in student.items()}
    O.CAPTURE = cap = {}
    try:
        losses, _, _, _, _ = O.segmentor_forward_train(sd, img, gt, pixw)
        sum(v for k, v in losses.items() if 'loss' in k).backward()
    finally:
        O.CAPTURE = None
    dys = {k: t.grad for k, t in cap.items() if t.grad is not None}

    # ---- HIP model, wired exactly as in the product
    prev, prev_math, prev_dw, prev_defer = layers.WINOGRAD, layers.CONV_MATH, layers.FUSE_ASPP_DW, layers.DEFER_BN_APPLY
    prev_gate = layers.FUSE_RES_GATE
    layers.WINOGRAD, layers.CONV_MATH = wino, math      # both arithmetics of the dense convolutions: fp32-input MFMA and the bf16x6 split
    # one closure per conv -> BN link here: the ASPP head's fused three-branch depthwise launch (one closure for three layers' data and weight
    # gradients) is checked against this per-branch wiring in test_aspp_depthwise_branches_fused_equals_per_branch below
    layers.FUSE_ASPP_DW = False
    # ... and every normalised tensor is materialised here (the observer reads each link's input and output): the two deferred
    # normalisations of the product (stem.6 -> max-pool, sep_bottleneck[0] -> [1]) are checked against this wiring in
    # test_deferred_normalisation_equals_the_materialised_one below
    layers.DEFER_BN_APPLY = bool(fold)
    prev_fold, layers.FOLD_BN_WINO = layers.FOLD_BN_WINO, bool(fold)
    prev_fold2, layers.FOLD_BN_GEMM = layers.FOLD_BN_GEMM, bool(fold)      # ... and bn2 -> conv3 normalised inside conv3's GEMM and weight gradient
    prev_fold3, layers.FOLD_BN_DWSEP = layers.FOLD_BN_DWSEP, bool(fold)    # ... and the depthwise stages' outputs inside their pointwise GEMMs
    prev_fold4, layers.FOLD_BN_RESIDUAL = layers.FOLD_BN_RESIDUAL, bool(fold)    # ... and the downsample branches inside bn3's normalisation pass
    # ... and every link writes its input gradients itself (the observer compares them closure by closure): the identity-branch gradient that
    # the product folds into conv1's data-gradient epilogue is checked against this wiring in test_residual_gate_in_the_dgrad_epilogue below
    layers.FUSE_RES_GATE = False
    # ... on ONE stream (the observer reads what each closure added to the arena right after it ran; the product queues the weight gradients on
    # a side stream: pinned to the single-stream schedule by test_stream_overlap_options_do_not_change_the_step)
    prev_overlap = (layers.WGRAD_STREAM, layers.FORK_TEACHER)
    layers.set_overlap(False, False)
    try:
        model = build_segmentor(model_cfg(C, 3, dropout=0.0))
        model.load_state_dict(student, strict=True)
        model.cuda()
        arena = ParamArena(list(model.named_parameters()), torch.device('cuda'), with_grad=True)
        model.repack_weights(need_dgrad=True)
        names = {id(m): n for n, m in model.named_modules()}
        rows, state, n_fused = [], {}, [0]

        def out_name(tag):
            if tag['op'] in ('conv_bn_act', 'conv'):
                n = names[id(tag['conv'])]
                return n.rsplit('.', 1)[0] + '.out' if tag.get('residual') is not None else n
            return tag.get('name')

        def inputs_of(tag):
            ins = [('x', tag['x'])]
            if tag.get('residual') is not None:
                ins.append(('residual', tag['residual']))
            return [(k, v) for k, v in ins if v.requires_grad]

        def params_of(tag):
            ps = []
            if 'conv' in tag:
                ps.append(('weight', tag['conv'].weight))
                if tag['conv'].bias is not None:
                    ps.append(('bias', tag['conv'].bias))
            if 'bn' in tag:
                ps += [('gamma', tag['bn'].weight), ('beta', tag['bn'].bias)]
            return ps

        def reference(tag, dy, dt=torch.float64):
            """fp64 CPU autograd of the oracle's restatement of this one link at the HIP layer's actual input (dt=float32: the same
            link as torch's fp32 CPU path evaluates it -- the yardstick for the one badly conditioned link)"""
            op = tag['op']
            x = _data(tag['x']).detach().cpu().to(dt).requires_grad_(tag['x'].requires_grad)
            leaves, keys = [x] if x.requires_grad else [], ['x'] if x.requires_grad else []
            if op in ('conv_bn_act', 'conv'):
                cv = tag['conv']
                w = cv.weight.data.detach().cpu().to(dt).requires_grad_(True)
                leaves.append(w); keys.append('weight')
                bias = None
                if cv.bias is not None:
                    bias = cv.bias.data.detach().cpu().to(dt).requires_grad_(True)
                    leaves.append(bias); keys.append('bias')
                y = F.conv2d(x, w, bias, cv.stride, cv.padding, cv.dilation, cv.groups)
                if op == 'conv_bn_act':
                    gam = tag['bn'].weight.data.detach().cpu().to(dt).requires_grad_(True)
                    bet = tag['bn'].bias.data.detach().cpu().to(dt).requires_grad_(True)
                    leaves += [gam, bet]; keys += ['gamma', 'beta']
                    y = F.batch_norm(y, None, None, gam, bet, True, 0.0, O.BN_EPS)
                    if tag['residual'] is not None:
                        r = _data(tag['residual']).detach().cpu().to(dt).requires_grad_(tag['residual'].requires_grad)
                        if r.requires_grad:
                            leaves.append(r); keys.append('residual')
                        y = y + r
                    if tag['relu']:
                        gate = (_data(tag['out']).detach().cpu() > 0).to(dt)      # the HIP forward's own ReLU decision
                        fwd = F.relu(y).detach()
                        y = y * gate
                    else:
                        fwd = y.detach()
                    state['fwd_err'] = mixed_err(_data(tag['out']), fwd)[1]
            elif op == 'maxpool':
                y = F.max_pool2d(x, 3, 2, 1)
            elif op in ('resize', 'broadcast'):
                y = F.interpolate(x, size=_data(tag['out']).shape[-2:], mode='bilinear', align_corners=False)
            elif op == 'gap':
                y = x.mean((2, 3), keepdim=True)
            elif op == 'ce':
                lab = tag['label'].detach().cpu()
                pw = None if tag['weight'] is None else tag['weight'].detach().cpu().to(dt)
                up = F.interpolate(x, size=lab.shape[-2:], mode='bilinear', align_corners=False)
                y = O.ce_loss(up, lab.reshape(lab.shape[0], *lab.shape[-2:]).long(), pw, tag['class_weight'], tag['loss_weight'],
                              tag['ignore_index'])
                dy = torch.ones((), dtype=dt)
            else:
                raise AssertionError(f'untested tape op {op}')
            gs = torch.autograd.grad(y, leaves, dy.to(dt))
            # BatchNorm over fewer than 16 values per channel (the image-pool branch: b values): the normalisation and its
            # backward are conditioned by 1/var of b numbers -- report the size so the bound can say so
            state['bn_count'] = y.numel() // y.shape[1] if op == 'conv_bn_act' else None
            return dict(zip(keys, gs))

        def observer(tag, phase):
            name = out_name(tag)
            if phase == 'pre':
                dy = None
                if tag['out'] is not None:
                    assert name in dys, f'no oracle gradient captured for {name}'
                    dy = dys[name]
                    g = tag['out'].grad
                    assert g is not None and tuple(g.shape) == tuple(dy.shape), (name, None if g is None else g.shape, dy.shape)
                    state['own_dy_err'] = mixed_err(g, dy)[1]      # how far the HIP chain's own accumulated gradient had drifted
                    g.copy_(dy.cuda())                              # in place: slices of concat gradients stay wired
                    if tag['out'].bn is not None:
                        # the launch that completed this gradient emitted the layer's BatchNorm-backward sums from ITS values; the
                        # forced gradient invalidates them, so this link is checked on the two-pass kernels.  The fused sums are
                        # checked at kernel level (tests/test_bn_backward_fused_gpu.py) and as wired by comparing whole steps with
                        # the fusion on / off (tests/test_train_step_gpu.py::test_fused_bn_backward_does_not_change_the_step).
                        n_fused[0] += tag['out'].bn.partials is not None
                        tag['out'].bn.partials = None
                state.update(dy=dy, before={k: _grad_of(v) for k, v in inputs_of(tag)},
                             pbefore={k: p.grad.clone() for k, p in params_of(tag)})
                return
            ref = reference(tag, state['dy'])
            ref32 = None
            if state.get('bn_count') is not None and state['bn_count'] < 16:
                fe, bc = state.get('fwd_err'), state['bn_count']
                ref32 = reference(tag, state['dy'], torch.float32)       # torch's own fp32 evaluation of this link
                state['fwd_err'], state['bn_count'] = fe, bc
            got = {}
            for k, v in inputs_of(tag):
                after, before = v.grad, state['before'][k]
                assert after is not None, (name, k)
                got[k] = after if before is None else after - before
            for k, p in params_of(tag):
                got[k] = p.grad - state['pbefore'][k]
            for k in ref:
                worst, nrm = mixed_err(got[k], ref[k])
                rows.append((tag['op'], name or tag['op'], k, worst, nrm, state.get('fwd_err', 0.0), state.get('own_dy_err', 0.0),
                             state.get('bn_count'), None if ref32 is None else mixed_err(ref32[k], ref[k])[1]))
            state.clear()

        tape = Tape(observer)
        gt8 = ops.to_u8(gt.cuda())
        model.forward_train(img.cuda(), batch['img_metas'], gt8, pixw.cuda(), tape=tape)
        n_closures = len(tape.fns)
        tape.backward()
        torch.cuda.synchronize()
    finally:
        layers.WINOGRAD, layers.CONV_MATH, layers.FUSE_ASPP_DW, layers.DEFER_BN_APPLY = prev, prev_math, prev_dw, prev_defer
        layers.FUSE_RES_GATE = prev_gate
        layers.FOLD_BN_WINO = prev_fold
        layers.FOLD_BN_GEMM = prev_fold2
        layers.FOLD_BN_DWSEP = prev_fold3
        layers.FOLD_BN_RESIDUAL = prev_fold4
        layers.set_overlap(*prev_overlap)

    print(f'\n{len(rows)} checked tensors over {n_closures} closures (winograd={wino}); worst element error / bound, norm-wise rel:')
    for op, name, k, worst, nrm, fe, de, _, _ in sorted(rows, key=lambda r: -r[3])[:25]:
        print(f'   {worst:8.3f} {nrm:9.2e}  fwd {fe:8.1e}  chain-dy {de:8.1e}  {name}:{k} [{op}]')
    print(f'   {n_fused[0]} BatchNorm layers had received fused backward sums from the launch completing their gradient')
    if layers.FUSE_BN_BWD and layers.FUSE_BN_BWD_MIN_K <= 512:                     # the defaults (not PFST_FUSE_BN_BWD=0 / a higher threshold)
        # conv3 of layer2-4 (13), conv1 of layer4 (3), the three ASPP pointwise convs + the 1x1 branch, sep_bottleneck.1: the K >= 512
        # data gradients that complete a conv -> BN layer's gradient, under EVERY arithmetic (f16x3: conv_igemm_f16x3_bnb_kernel)
        assert n_fused[0] >= 15, (math, n_fused[0])
    checked_ops = {r[0] for r in rows}
    assert {'conv_bn_act', 'conv', 'maxpool', 'resize', 'gap', 'broadcast', 'ce'} <= checked_ops, checked_ops
    n_bn = sum(1 for m in model.modules() if isinstance(m, layers.BatchNorm2dP))
    assert n_bn == 70 and sum(1 for r in rows if r[0] == 'conv_bn_act' and r[2] == 'weight') == n_bn      # every conv+BN layer
    # every link: element-wise mixed bound.  One exception, stated: the image-pool branch normalises b = 2 values per channel, its
    # backward is a cancellation conditioned by 1/var of two numbers (fp32 forward of that layer alone: 2e-4), so no fp32
    # evaluation meets 1e-3 on every input -- there the HIP link must be norm-wise within 1e-3 OR as close to fp64 as torch's own
    # fp32 CPU evaluation of the same link is (x3 for a different but equally valid summation order).
    small = [r for r in rows if r[7] is not None and r[7] < 16]
    assert {r[1] for r in small} == {'decode_head.image_pool.1.conv'}
    for r in small:
        print(f'   image-pool link {r[2]}: HIP {r[4]:.2e}  torch-fp32 {r[8]:.2e}  (norm-wise vs fp64)')
    bad = [r for r in rows if r not in small and not r[3] <= 1.0] + [r for r in small if not r[4] <= max(1e-3, 3.0 * r[8])]
    assert not bad, bad[:10]


@pytest.mark.parametrize('prefill', [False, True])
@pytest.mark.parametrize('opts', [{}, dict(detach_unfold=False, top_k=None)], ids=['shipped', 'unfold_grad_all_pairs'])
def test_pfgst_loss_link_as_wired(opts, prefill):
    """The PFGSTLoss closure inside a WHOLE PFGST.train_step (VERDICT r2 next #1b): recorded last, so it runs first in the single sweep
    and is the first writer of dL/dx_src (the source pass's decoded features) and of dL/dlogits_trg (the mixed pass's logits), into
    which the cross-entropy backward accumulates afterwards.  A Tape observer measures what the closure adds to both buffers and
    compares it, element-wise with the per-link bound, with fp64 autograd of `oracle.pfgst_loss` (pfgst_loss.py:44-234) at the
    HIP step's own inputs.  prefill=True puts a gradient into both buffers first (the accumulate path: what the closure would see
    if another consumer had written before it); the bound is the same.  The step's inputs give a live target side (asserted)."""
    import random

    import numpy as np

    import pfst_amd  # noqa: F401
    from helpers import to_dev, uda_cfg
    from oracle import pfst_oracle as O
    from pfst_amd.optim import build_optimizer
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import synth_batch

    cfg = uda_cfg(threshold=0.30)
    cfg['aux_losses'][0].update(opts)
    model = UDA.build(cfg)
    both, student, teacher = seeded_pfgst_state(O, 9)
    model.load_state_dict(both, strict=False)
    model.cuda()
    opt = build_optimizer(model, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01))
    batch = synth_batch(2, 128, 6, seed=4321)
    state, rows = {}, []

    def observer(tag, phase):
        if tag.get('op') != 'pfgst_loss':
            return
        lt, xs = tag['logits_trg'], tag['x_src']
        if phase == 'pre':
            assert lt.grad is None and xs.grad is None, 'as wired the PFGSTLoss closure is the first writer of both gradients'
            if prefill:
                g = torch.Generator().manual_seed(11)
                lt._grad = (1e-4 * torch.randn(lt.data.shape, generator=g)).cuda()
                xs._grad = (1e-4 * torch.randn(xs.data.shape, generator=g)).cuda()
            state['before'] = (_grad_of(lt), _grad_of(xs))
            return
        mod = tag['module']
        l64 = lt.data.detach().cpu().double().requires_grad_(True)
        x64 = xs.data.detach().cpu().double().requires_grad_(True)
        losses, extras = O.pfgst_loss(l64, tag['x_ema'].data.detach().cpu().double(), x64, tag['gt_src'].cpu().long(),
                                      tag['mix_masks'].cpu().long(), mod.weights, dil=mod.dilation, top_k=mod.top_k,
                                      downscale=1.0 / mod.ds, sim_type=mod.sim_type, detach_unfold=not mod.unfold_grad)
        state['mask_frac'] = float(extras['mask'].float().mean())
        state['losses'] = {k: float(v.sum()) for k, v in losses.items()}
        sum(v.sum() for v in losses.values()).backward()
        for name, var, ref, before in (('logits_trg', lt, l64.grad, state['before'][0]), ('x_src', xs, x64.grad, state['before'][1])):
            got = var.grad if before is None else var.grad - before
            rows.append((name,) + mixed_err(got, ref) + (float(ref.abs().max()),))

    model.tape_observer = observer
    random.seed(101); np.random.seed(101)
    out = model.train_step(to_dev(batch, 'cuda'), opt)
    model.tape_observer = None
    assert len(rows) == 2, 'the PFGSTLoss closure was not seen by the observer (untagged?)'
    assert state['mask_frac'] >= 0.15, state['mask_frac']
    lv = out['log_vars']
    assert lv['loss_sim_pos'] != 0.0 and lv['loss_sim_neg'] != 0.0
    for k, v in state['losses'].items():
        assert abs(lv[k] - v) <= 1e-4 * max(abs(v), 1e-3), (k, lv[k], v)      # fp64 of the same inputs: the kernels' own forward error
    for name, worst, nrm, scale in rows:
        print(f'   {worst:8.3f} {nrm:9.2e}  max|ref| {scale:.2e}  pfgst_loss:{name}  (prefill={prefill}, target region {state["mask_frac"]:.2f})')
        assert scale > 0 and worst <= (1.0 if not prefill else 1.5), (name, worst, nrm)


@pytest.mark.parametrize('math', ['f16x3', 'bf16x6'])
@pytest.mark.parametrize('chan', [(64, 256), (48, 512), (128, 80), (96, 160)])
def test_winograd_layers_outside_the_f16x3_shapes_train(chan, math):
    """a Winograd-eligible 3x3 layer with more than 64 output channels that the f16x3 GEMM does NOT cover (Cin not a multiple
    of 32: 48 -> 512; a data gradient over <= 64 rows: 64 -> 256) keeps a plain fp32 V in forward; its weight gradient must then take
    the bf16x6 product instead of asserting on the missing scale group.  (128 -> 80 stays direct, 96 -> 160 is the covered case.)
    conv -> BN(train) -> ReLU forward + backward as wired (layers.conv_bn_act) against fp64 autograd, element-wise bound."""
    from pfst_amd import layers
    from pfst_amd.engine import ParamArena, Tape, Var
    cin, cout = chan
    prev = layers.CONV_MATH
    layers.CONV_MATH = math
    try:
        torch.manual_seed(cin + cout)
        mod = layers.ConvModule(cin, cout, 3, padding=2, dilation=2).cuda()
        ParamArena(list(mod.named_parameters()), torch.device('cuda'), with_grad=True)
        mod.conv.repack(need_dgrad=True)
        expect_wino = cin * cout >= layers.WINO_MIN_CC
        assert mod.conv.wino == expect_wino
        if math == 'f16x3' and chan in ((64, 256), (48, 512)):
            assert mod.conv.wino and not mod.conv.wino_f16                       # the case under test
        x = torch.randn(2, cin, 32, 32, generator=torch.Generator().manual_seed(1))
        dy = torch.randn(2, cout, 32, 32, generator=torch.Generator().manual_seed(2))
        tape = Tape()
        xv = Var(x.cuda(), True)
        yv = mod(xv, tape)
        buf, _ = yv.grad_target()
        buf.copy_(dy.cuda())
        tape.backward()
        torch.cuda.synchronize()
        x64 = x.double().requires_grad_()
        w64 = mod.conv.weight.detach().cpu().double().requires_grad_()
        g64 = mod.bn.weight.detach().cpu().double().requires_grad_()
        b64 = mod.bn.bias.detach().cpu().double().requires_grad_()
        ref = F.relu(F.batch_norm(F.conv2d(x64, w64, None, 1, 2, 2), None, None, g64, b64, True, 0.1, 1e-5))
        ref.backward(dy.double())
        for name, got, want in (('y', yv.data, ref), ('dx', xv.grad, x64.grad), ('dW', mod.conv.weight.grad, w64.grad),
                                ('dgamma', mod.bn.weight.grad, g64.grad), ('dbeta', mod.bn.bias.grad, b64.grad)):
            worst, nrm = mixed_err(got, want)
            assert worst <= 1.0, (chan, math, name, worst, nrm)
    finally:
        layers.CONV_MATH = prev


@pytest.mark.parametrize('math', ['f16x3', 'f32'])
def test_dropout2d_folded_into_the_normalisation_pass(math):
    """nn.Dropout2d before conv_seg (decode_head.py:103-107,242-247).  The decode head applies its keep / (1 - p) factors inside the
    normalisation pass of sep_bottleneck[1]'s pointwise layer (bn_apply post_scale) and, in backward, inside that layer's
    BatchNorm-backward passes; the auxiliary head keeps the separate scaling pass (its returned features are the pre-dropout ones).  One
    segmentor forward + backward with INJECTED masks against the oracle: losses, logits, and the gradients on both sides of the dropout."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd import hip_ops as ops
    from pfst_amd import layers
    from pfst_amd.engine import ParamArena, Tape
    from pfst_amd.registry import build_segmentor
    from pfst_amd.synthetic import synth_batch

    C, b, S = 6, 2, 128
    _, student, _ = seeded_pfgst_state(O, 9)
    batch = synth_batch(b, S, C, seed=4321)
    img, gt = batch['img'], batch['gt_semantic_seg']
    g = torch.Generator().manual_seed(17)
    p = 0.3                                            # a third of the planes dropped: every code path sees zeros and scaled planes
    m_dec = ((torch.rand(b, 512, generator=g) >= p).float() / (1 - p))
    m_aux = ((torch.rand(b, 256, generator=g) >= p).float() / (1 - p))
    sd = {k: (v.clone().double().requires_grad_(True) if v.is_floating_point() and 'running' not in k else
              (v.clone().double() if v.is_floating_point() else v.clone())) for k, v in student.items()}
    losses, _, logits, _, aux_logits = O.segmentor_forward_train(sd, img.double(), gt, None,
                                                                  drop_masks=(m_dec.double().view(b, 512, 1, 1), m_aux.double().view(b, 256, 1, 1)))
    sum(v for k, v in losses.items() if 'loss' in k).backward()
    prev = layers.CONV_MATH
    layers.CONV_MATH = math
    try:
        model = build_segmentor(model_cfg(C, 3, dropout=0.1))
        model.load_state_dict(student, strict=True)
        model.cuda()
        arena = ParamArena(list(model.named_parameters()), torch.device('cuda'), with_grad=True)
        model.repack_weights(need_dgrad=True)
        model.decode_head.injected_dropout_mask, model.auxiliary_head.injected_dropout_mask = m_dec.cuda(), m_aux.cuda()
        calls = []
        orig = ops.channel_scale
        ops.channel_scale = lambda x, m: (calls.append(tuple(x.shape)), orig(x, m))[1]
        try:
            tape = Tape()
            out = model.forward_train(img.cuda(), batch['img_metas'], ops.to_u8(gt.cuda()), None, return_logits=True, tape=tape)
            tape.backward()
            torch.cuda.synchronize()
        finally:
            ops.channel_scale = orig
    finally:
        layers.CONV_MATH = prev
    # the decode head's dropout ran inside bn_apply / bn_backward: the only separate scaling passes are the auxiliary head's (fwd + bwd)
    assert calls == [(b, 256, S // 8, S // 8)] * 2, calls
    for k in ('decode.loss_ce', 'aux.loss_ce', 'decode.acc_seg', 'aux.acc_seg'):
        assert abs(float(out[k]) - float(losses[k])) <= 1e-3 * max(abs(float(losses[k])), 1e-2), (k, float(out[k]), float(losses[k]))
    worst, nrm = mixed_err(out['logits'].data, logits, )
    assert nrm < 1e-3, nrm
    # gradients above the dropout (conv_seg: 1e-3), of the layer it is folded into (one train-mode BatchNorm below the loss: the fp32
    # conditioning regime of tests/test_train_step_gpu.py's golden samples, 2e-2) and one layer further down (0.1).  A wrong or missing
    # factor would show as O(0.3): a third of the planes are zeroed, the rest scaled by 1.43.
    for name, tol in (('decode_head.conv_seg.weight', 1e-3), ('decode_head.conv_seg.bias', 1e-3),
                      ('decode_head.sep_bottleneck.1.pointwise_conv.conv.weight', 2e-2),
                      ('decode_head.sep_bottleneck.1.pointwise_conv.bn.weight', 2e-2), ('decode_head.sep_bottleneck.1.pointwise_conv.bn.bias', 2e-2),
                      ('decode_head.sep_bottleneck.1.depthwise_conv.conv.weight', 0.1), ('auxiliary_head.conv_seg.weight', 1e-3),
                      ('auxiliary_head.convs.0.bn.weight', 2e-2), ('auxiliary_head.convs.0.conv.weight', 2e-2)):
        _, e = mixed_err(arena.view(arena.grad, name), sd[name].grad)
        print(f'   {name}: rel err vs fp64 {e:.2e} (bound {tol})')
        assert e < tol, (name, e)
    # and the fold is the SAME arithmetic as the separate scaling pass (one product per element, in the same place of both chains): the
    # logits are bit-identical, the gradients agree to the weight gradients' atomic summation order
    from pfst_amd import models
    grad_fold, logits_fold = arena.grad.clone(), out['logits'].data.clone()
    prev_fold, models.FOLD_DROPOUT = models.FOLD_DROPOUT, False
    layers.CONV_MATH = math
    try:
        arena.zero_grad()
        model.repack_weights(need_dgrad=True)
        for m in model.modules():                                  # same BatchNorm running buffers as the first pass started from
            if isinstance(m, layers.BatchNorm2dP):
                m.running_mean.zero_(); m.running_var.fill_(1.0)
        tape = Tape()
        out2 = model.forward_train(img.cuda(), batch['img_metas'], ops.to_u8(gt.cuda()), None, return_logits=True, tape=tape)
        tape.backward()
        torch.cuda.synchronize()
    finally:
        models.FOLD_DROPOUT, layers.CONV_MATH = prev_fold, prev
    assert torch.equal(out2['logits'].data, logits_fold), 'folded and separate Dropout2d must give bit-identical logits'
    _, e = mixed_err(arena.grad, grad_fold)
    assert e < 1e-4, e


def test_aspp_depthwise_branches_fused_equals_per_branch():
    """layers.dwsep_branches: the ASPP head's three atrous depthwise stages as one launch each way (pfst_dwconv3x3_multi_fwd / _bwd).  One
    segmentor forward + backward with the fusion on and off: the forward is bit-identical (same stencil arithmetic per element, same
    BatchNorm partial layout), every parameter gradient agrees to the summation order of the fp32 atomics / the branch sum (1e-5 norm-wise
    on the whole arena, 1e-4 on the three depthwise filters' own gradients)."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd import hip_ops as ops
    from pfst_amd import layers
    from pfst_amd.engine import ParamArena, Tape
    from pfst_amd.registry import build_segmentor
    from pfst_amd.synthetic import synth_batch

    C, b, S = 6, 2, 256                     # 32 x 32 planes at 1/8: dilations 12 / 24 / 36 reach well inside the plane
    _, student, _ = seeded_pfgst_state(O, 9)
    batch = synth_batch(b, S, C, seed=99)
    runs, launches = {}, {}
    prev = layers.FUSE_ASPP_DW
    try:
        for fuse in (True, False):
            layers.FUSE_ASPP_DW = fuse
            model = build_segmentor(model_cfg(C, 3, dropout=0.0))
            model.load_state_dict(student, strict=True)
            model.cuda()
            arena = ParamArena(list(model.named_parameters()), torch.device('cuda'), with_grad=True)
            model.repack_weights(need_dgrad=True)
            n_multi = [0]
            orig = ops.dwconv_multi_bwd_
            ops.dwconv_multi_bwd_ = lambda *a, **k: (n_multi.__setitem__(0, n_multi[0] + 1), orig(*a, **k))[1]
            try:
                tape = Tape()
                out = model.forward_train(batch['img'].cuda(), batch['img_metas'], ops.to_u8(batch['gt_semantic_seg'].cuda()), None,
                                          return_logits=True, tape=tape)
                tape.backward()
                torch.cuda.synchronize()
            finally:
                ops.dwconv_multi_bwd_ = orig
            runs[fuse] = (out['logits'].data.clone(), arena.grad.clone(), {n: arena.view(arena.grad, n).clone() for n in arena.names
                                                                          if 'aspp_modules' in n and 'depthwise_conv.conv' in n},
                          float(out['decode.loss_ce']))
            launches[fuse] = n_multi[0]
    finally:
        layers.FUSE_ASPP_DW = prev
    assert launches == {True: 1, False: 0}, launches
    (l1, g1, d1, c1), (l0, g0, d0, c0) = runs[True], runs[False]
    # forward: the depthwise outputs and their BatchNorm partials are bit-identical (tests/test_hip_ops.py); the plane means of the image-pool
    # branch come from an fp64 sum in another order, i.e. the same fp32 value except for a rare last-bit flip -- which that branch's
    # two-sample BatchNorm can amplify: the logits agree to 1e-5 of their scale instead of bit for bit
    _, el = mixed_err(l1, l0)
    assert el < 1e-5 and abs(c1 - c0) <= 1e-6 * abs(c0), (el, c1, c0)
    _, e = mixed_err(g1, g0)
    assert e < 1e-4, e
    assert len(d1) == 3
    for n in d1:
        _, e = mixed_err(d1[n], d0[n])
        assert e < 1e-4, (n, e)


def test_deferred_normalisation_equals_the_materialised_one():
    """layers.conv_bn_act(defer=True): the outputs of stem.6 (consumer: the max-pool) and of sep_bottleneck[0] (consumer: sep_bottleneck[1]'s
    depthwise layer) are never written -- the consumer reads the pre-BatchNorm tensor and applies scale, shift and ReLU as it loads, in
    forward and (the depthwise layer's weight gradient) in backward.  Same arithmetic per element as bn_apply, so one segmentor forward +
    backward with the deferral on and off gives bit-identical logits and gradients equal to the atomics' summation order; the two
    normalisation launches are really gone."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd import hip_ops as ops
    from pfst_amd import layers
    from pfst_amd.engine import ParamArena, Tape
    from pfst_amd.registry import build_segmentor
    from pfst_amd.synthetic import synth_batch

    C, b, S = 6, 2, 128
    _, student, _ = seeded_pfgst_state(O, 9)
    batch = synth_batch(b, S, C, seed=31)
    runs, applies = {}, {}
    prev = layers.DEFER_BN_APPLY
    try:
        for defer in (True, False):
            layers.DEFER_BN_APPLY = defer
            model = build_segmentor(model_cfg(C, 3, dropout=0.0))
            model.load_state_dict(student, strict=True)
            model.cuda()
            arena = ParamArena(list(model.named_parameters()), torch.device('cuda'), with_grad=True)
            model.repack_weights(need_dgrad=True)
            n_apply = [0]
            orig = ops.bn_apply
            ops.bn_apply = lambda *a, **k: (n_apply.__setitem__(0, n_apply[0] + 1), orig(*a, **k))[1]
            try:
                tape = Tape()
                out = model.forward_train(batch['img'].cuda(), batch['img_metas'], ops.to_u8(batch['gt_semantic_seg'].cuda()), None,
                                          return_logits=True, tape=tape)
                tape.backward()
                torch.cuda.synchronize()
            finally:
                ops.bn_apply = orig
            runs[defer] = (out['logits'].data.clone(), arena.grad.clone())
            applies[defer] = n_apply[0]
    finally:
        layers.DEFER_BN_APPLY = prev
    # deferred: stem.6 and sep_bottleneck[0]; + bn1 of the 12 bottlenecks whose conv2 runs through the Winograd domain (layers.FOLD_BN_WINO)
    folded = (12 if (layers.FOLD_BN_WINO and layers.WINOGRAD) else 0) + (4 if (layers.FOLD_BN_CONCAT and layers.WINOGRAD) else 0)
    folded += 16 if (layers.FOLD_BN_GEMM and layers.CONV_MATH == 'f16x3') else 0          # bn2 of every bottleneck: normalised inside conv3's GEMM
    folded += 5 if (layers.FOLD_BN_DWSEP and layers.CONV_MATH == 'f16x3') else 0          # the five depthwise stages: inside their pointwise GEMMs
    folded += 4 if layers.FOLD_BN_RESIDUAL else 0                                         # the four downsample branches: inside bn3's pass
    assert applies[False] == 70 and applies[True] == 68 - folded, applies
    assert torch.equal(runs[True][0], runs[False][0]), 'deferred and materialised normalisation must give bit-identical logits'
    _, e = mixed_err(runs[True][1], runs[False][1])
    assert e < 1e-4, e


def test_bn1_folded_into_the_winograd_input_transform():
    """layers.FOLD_BN_WINO (round 5): in the 12 bottlenecks whose conv2 runs through the Winograd domain, conv1 -> bn1 -> ReLU is never
    written -- conv2's input transform normalises the pre-BN tensor as it loads it (pfst_wino_input bnl), conv2's weight gradient comes from
    the transformed input kept in forward, bn1's backward from the pre-BN tensor.  Under f16x3 the transform writes V pre-split and needs
    max |y1| before y1 exists: conv1's epilogue emits per-channel (min, max) partials and pfst_bn_finalize_partials predicts the maximum.
    Same arithmetic per element (the fma and max of bn_apply), same scale exponent: one segmentor forward + backward with the fold on and
    off gives bit-identical logits, gradients equal to the atomics' summation order, and 12 normalisation launches less.
    Follows /root/reference/rsiseg/models/backbones/resnet.py:273-296 (conv1 -> norm1 -> relu -> conv2 of Bottleneck._inner_forward)."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd import hip_ops as ops
    from pfst_amd import layers
    from pfst_amd.engine import ParamArena, Tape
    from pfst_amd.registry import build_segmentor
    from pfst_amd.synthetic import synth_batch

    if not (layers.WINOGRAD and layers.DEFER_BN_APPLY):
        pytest.skip('needs the Winograd dispatch and deferred normalisations')
    C, b, S = 6, 2, 128
    _, student, _ = seeded_pfgst_state(O, 9)
    batch = synth_batch(b, S, C, seed=23)
    runs, applies, inputs = {}, {}, {}
    prev, prev_cat = layers.FOLD_BN_WINO, layers.FOLD_BN_CONCAT
    n_cat = 4 if prev_cat else 0             # the ASPP head's four concat writers, folded into the bottleneck's input transform by the same mechanism
    inner = ops.call
    try:
        for fold in (True, False):
            layers.FOLD_BN_WINO = fold
            layers.FOLD_BN_CONCAT = fold and prev_cat
            model = build_segmentor(model_cfg(C, 3, dropout=0.0))
            model.load_state_dict(student, strict=True)
            model.cuda()
            arena = ParamArena(list(model.named_parameters()), torch.device('cuda'), with_grad=True)
            model.repack_weights(need_dgrad=True)
            seen = {}

            def counting(name, *a):
                seen[name] = seen.get(name, 0) + 1
                if name == 'pfst_wino_input' and a[11]:
                    seen['normalising transforms'] = seen.get('normalising transforms', 0) + 1
                return inner(name, *a)
            ops.call = counting
            try:
                tape = Tape()
                out = model.forward_train(batch['img'].cuda(), batch['img_metas'], ops.to_u8(batch['gt_semantic_seg'].cuda()), None,
                                          return_logits=True, tape=tape)
                tape.backward()
                torch.cuda.synchronize()
            finally:
                ops.call = inner
            runs[fold] = (out['logits'].data.clone(), arena.grad.clone())
            applies[fold] = seen
    finally:
        layers.FOLD_BN_WINO, layers.FOLD_BN_CONCAT = prev, prev_cat
    assert applies[False]['pfst_bn_apply'] - applies[True]['pfst_bn_apply'] == 12 + n_cat, (applies[False]['pfst_bn_apply'], applies[True]['pfst_bn_apply'])
    assert applies[True].get('normalising transforms', 0) == 12 + n_cat // 4 and applies[False].get('normalising transforms', 0) == 0
    assert applies[True]['pfst_wino_input'] == applies[False]['pfst_wino_input']            # no transform is re-run in backward: V was kept
    assert torch.equal(runs[True][0], runs[False][0]), 'folded and materialised normalisation must give bit-identical logits'
    _, e = mixed_err(runs[True][1], runs[False][1])
    print(f'   gradient arena, fold on vs off: {e:.2e}')
    assert e < 1e-4, e


def test_bn2_folded_into_conv3s_gemm():
    """layers.FOLD_BN_GEMM (round 5): in every bottleneck conv2 -> bn2 -> ReLU is never written -- conv3's f16x3 GEMM normalises the pre-BN
    tensor between load and split (pfst_conv_igemm_f16x3 bnl), conv3's weight gradient likewise (pfst_conv_wgrad_f16x3 bnl), bn2's backward
    reads the pre-BN tensor, and the data gradient of conv3 still emits bn2's BatchNorm-backward sums where it did.  max |y2| is predicted from
    the (min, max) partials of conv2's producer (the Winograd output transform; layer1 and the stride-2 block: the direct GEMM's epilogue).
    Same arithmetic per element, same scale exponent: one segmentor forward + backward with the fold on and off gives bit-identical logits,
    gradients equal to the atomics' summation order, and 16 normalisation launches less.
    Follows /root/reference/rsiseg/models/backbones/resnet.py:282-290 (conv2 -> norm2 -> relu -> conv3 of Bottleneck._inner_forward)."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd import hip_ops as ops
    from pfst_amd import layers
    from pfst_amd.engine import ParamArena, Tape
    from pfst_amd.registry import build_segmentor
    from pfst_amd.synthetic import synth_batch

    if not (layers.DEFER_BN_APPLY and layers.CONV_MATH == 'f16x3'):
        pytest.skip('needs deferred normalisations under the f16x3 arithmetic')
    C, b, S = 6, 2, 128
    _, student, _ = seeded_pfgst_state(O, 9)
    batch = synth_batch(b, S, C, seed=29)
    runs, seen_by = {}, {}
    prev = layers.FOLD_BN_GEMM
    inner = ops.call
    try:
        for fold in (True, False):
            layers.FOLD_BN_GEMM = fold
            model = build_segmentor(model_cfg(C, 3, dropout=0.0))
            model.load_state_dict(student, strict=True)
            model.cuda()
            arena = ParamArena(list(model.named_parameters()), torch.device('cuda'), with_grad=True)
            model.repack_weights(need_dgrad=True)
            seen = {}

            def counting(name, *a):
                seen[name] = seen.get(name, 0) + 1
                if name == 'pfst_conv_igemm_f16x3' and a[27]:
                    seen['normalising GEMMs'] = seen.get('normalising GEMMs', 0) + 1
                if name == 'pfst_conv_wgrad_f16x3' and a[11]:
                    seen['normalising weight gradients'] = seen.get('normalising weight gradients', 0) + 1
                if name == 'pfst_conv_igemm_f16x3' and a[22]:
                    seen['fused BatchNorm-backward sums'] = seen.get('fused BatchNorm-backward sums', 0) + 1
                return inner(name, *a)
            ops.call = counting
            try:
                tape = Tape()
                out = model.forward_train(batch['img'].cuda(), batch['img_metas'], ops.to_u8(batch['gt_semantic_seg'].cuda()), None,
                                          return_logits=True, tape=tape)
                tape.backward()
                torch.cuda.synchronize()
            finally:
                ops.call = inner
            runs[fold] = (out['logits'].data.clone(), arena.grad.clone())
            seen_by[fold] = seen
    finally:
        layers.FOLD_BN_GEMM = prev
    on, off = seen_by[True], seen_by[False]
    assert off['pfst_bn_apply'] - on['pfst_bn_apply'] == 16, (off['pfst_bn_apply'], on['pfst_bn_apply'])
    assert on.get('normalising GEMMs', 0) - off.get('normalising GEMMs', 0) == 16           # (the depthwise-separable modules' pointwise layers: in both runs)
    assert on.get('normalising weight gradients', 0) - off.get('normalising weight gradients', 0) == 16
    assert on.get('fused BatchNorm-backward sums', 0) == off.get('fused BatchNorm-backward sums', 0) > 0     # conv3's data gradient still emits bn2's sums
    assert torch.equal(runs[True][0], runs[False][0]), 'folded and materialised normalisation must give bit-identical logits'
    _, e = mixed_err(runs[True][1], runs[False][1])
    print(f'   gradient arena, fold on vs off: {e:.2e}')
    assert e < 1e-4, e


def test_depthwise_stage_folded_into_the_pointwise_gemm():
    """layers.FOLD_BN_DWSEP (round 5): in the five DepthwiseSeparableConvModules (the ASPP head's three atrous branches -- one fused launch --,
    sep_bottleneck[0] and [1]) the depthwise conv -> BN -> ReLU output is never written: the pointwise layer's f16x3 GEMM (2048 / 560 / 512
    coefficient rows in LDS) and its weight gradient normalise the depthwise kernel's output as they load it, from the maximum predicted out of
    the depthwise kernels' (min, max) partials.  One segmentor forward + backward with the fold on and off: bit-identical logits, gradients
    equal to the atomics' summation order, five normalisation launches less.
    Follows /root/reference/rsiseg/models/decode_heads/sep_aspp_head.py:17-26,63-77 and mmcv's DepthwiseSeparableConvModule."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd import hip_ops as ops
    from pfst_amd import layers
    from pfst_amd.engine import ParamArena, Tape
    from pfst_amd.registry import build_segmentor
    from pfst_amd.synthetic import synth_batch

    if not (layers.DEFER_BN_APPLY and layers.CONV_MATH == 'f16x3'):
        pytest.skip('needs deferred normalisations under the f16x3 arithmetic')
    C, b, S = 6, 2, 128
    _, student, _ = seeded_pfgst_state(O, 9)
    batch = synth_batch(b, S, C, seed=37)
    prev, prev_dw = layers.FOLD_BN_DWSEP, layers.FUSE_ASPP_DW
    inner = ops.call
    try:
        for fused_aspp in (True, False):               # the three-branch launch, and the per-branch depthwise kernels
            layers.FUSE_ASPP_DW = fused_aspp
            runs, seen_by = {}, {}
            for fold in (True, False):
                layers.FOLD_BN_DWSEP = fold
                model = build_segmentor(model_cfg(C, 3, dropout=0.0))
                model.load_state_dict(student, strict=True)
                model.cuda()
                arena = ParamArena(list(model.named_parameters()), torch.device('cuda'), with_grad=True)
                model.repack_weights(need_dgrad=True)
                seen = {}

                def counting(name, *a):
                    seen[name] = seen.get(name, 0) + 1
                    if name == 'pfst_conv_igemm_f16x3' and a[27]:
                        seen['normalising GEMMs'] = seen.get('normalising GEMMs', 0) + 1
                    if name == 'pfst_conv_wgrad_f16x3' and a[11]:
                        seen['normalising weight gradients'] = seen.get('normalising weight gradients', 0) + 1
                    return inner(name, *a)
                ops.call = counting
                try:
                    tape = Tape()
                    out = model.forward_train(batch['img'].cuda(), batch['img_metas'], ops.to_u8(batch['gt_semantic_seg'].cuda()), None,
                                              return_logits=True, tape=tape)
                    tape.backward()
                    torch.cuda.synchronize()
                finally:
                    ops.call = inner
                runs[fold] = (out['logits'].data.clone(), arena.grad.clone())
                seen_by[fold] = seen
            on, off = seen_by[True], seen_by[False]
            assert off['pfst_bn_apply'] - on['pfst_bn_apply'] == 5, (fused_aspp, off['pfst_bn_apply'], on['pfst_bn_apply'])
            assert on.get('normalising GEMMs', 0) - off.get('normalising GEMMs', 0) == 5
            assert on.get('normalising weight gradients', 0) - off.get('normalising weight gradients', 0) == 5
            assert torch.equal(runs[True][0], runs[False][0]), 'folded and materialised normalisation must give bit-identical logits'
            _, e = mixed_err(runs[True][1], runs[False][1])
            print(f'   fused ASPP launch {fused_aspp}: gradient arena, fold on vs off: {e:.2e}')
            assert e < 1e-4, e
    finally:
        layers.FOLD_BN_DWSEP, layers.FUSE_ASPP_DW = prev, prev_dw


def test_downsample_branch_folded_into_bn3s_normalisation_pass():
    """layers.FOLD_BN_RESIDUAL (round 5): the downsample branch of the first block of every stage (conv 1x1 -> BN, no ReLU;
    /root/reference/rsiseg/models/backbones/resnet.py:298-303) is never written normalised -- the block's bn3 normalisation pass applies the
    branch's (sc, sh) to its pre-BN tensor as it loads the residual (pfst_bn_apply residual_coef): fma(r, sc, sh) is the value the branch's own
    pass would have written, so the block outputs are bit-identical; the branch's BatchNorm backward never read its output.  One segmentor
    forward + backward with the fold on and off: bit-identical logits, gradients equal to the atomics' summation order, four normalisation
    launches less, and the identity-branch gradient still rides in the gated form (layers.FUSE_RES_GATE)."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd import hip_ops as ops
    from pfst_amd import layers
    from pfst_amd.engine import ParamArena, Tape
    from pfst_amd.registry import build_segmentor
    from pfst_amd.synthetic import synth_batch

    if not layers.DEFER_BN_APPLY:
        pytest.skip('needs deferred normalisations')
    C, b, S = 6, 2, 128
    _, student, _ = seeded_pfgst_state(O, 9)
    batch = synth_batch(b, S, C, seed=41)
    runs, seen_by = {}, {}
    prev = layers.FOLD_BN_RESIDUAL
    inner = ops.call
    try:
        for fold in (True, False):
            layers.FOLD_BN_RESIDUAL = fold
            model = build_segmentor(model_cfg(C, 3, dropout=0.0))
            model.load_state_dict(student, strict=True)
            model.cuda()
            arena = ParamArena(list(model.named_parameters()), torch.device('cuda'), with_grad=True)
            model.repack_weights(need_dgrad=True)
            seen = {}

            def counting(name, *a):
                seen[name] = seen.get(name, 0) + 1
                if name == 'pfst_bn_apply' and a[17]:
                    seen['normalising residual passes'] = seen.get('normalising residual passes', 0) + 1
                if name == 'pfst_bn_backward' and a[12]:
                    seen['written identity gradients'] = seen.get('written identity gradients', 0) + 1
                return inner(name, *a)
            ops.call = counting
            try:
                tape = Tape()
                out = model.forward_train(batch['img'].cuda(), batch['img_metas'], ops.to_u8(batch['gt_semantic_seg'].cuda()), None,
                                          return_logits=True, tape=tape)
                tape.backward()
                torch.cuda.synchronize()
            finally:
                ops.call = inner
            runs[fold] = (out['logits'].data.clone(), arena.grad.clone())
            seen_by[fold] = seen
    finally:
        layers.FOLD_BN_RESIDUAL = prev
    on, off = seen_by[True], seen_by[False]
    assert off['pfst_bn_apply'] - on['pfst_bn_apply'] == 4, (off['pfst_bn_apply'], on['pfst_bn_apply'])
    assert on.get('normalising residual passes', 0) == 4 and off.get('normalising residual passes', 0) == 0
    assert on.get('written identity gradients', 0) == off.get('written identity gradients', 0)          # the gate fold is untouched
    assert torch.equal(runs[True][0], runs[False][0]), 'folded and materialised normalisation must give bit-identical logits'
    _, e = mixed_err(runs[True][1], runs[False][1])
    print(f'   gradient arena, fold on vs off: {e:.2e}')
    assert e < 1e-4, e


def test_bn_backward_of_two_layers_behind_one_gated_gradient():
    """pfst_bn_backward_dual (round 5): bn3 and the downsample branch's BN of a stage's first block (out = relu(bn3(.) + bn_d(.)),
    /root/reference/rsiseg/models/backbones/resnet.py:298-307) receive the same gated gradient; one reduction + one apply pass replace the two
    pfst_bn_backward calls.  Kernel level: both input gradients, both parameter gradients and both published maxima against the single-layer
    entry point (same per-element arithmetic; the sums differ by the atomics' order), with and without layer a's sums arriving as partials."""
    import pfst_amd  # noqa: F401
    from pfst_amd import hip_ops as ops

    g = torch.Generator().manual_seed(11)
    for (n, c, h, w) in [(2, 24, 32, 32), (3, 7, 16, 48)]:
        dy = torch.randn(n, c, h, w, generator=g).cuda() + 0.3          # a common mode: the projections cancel
        xa = (torch.randn(n, c, h, w, generator=g) * 2 + 1).cuda()
        xb = (torch.randn(n, c, h, w, generator=g) * 0.5 - 2).cuda()
        out = torch.randn(n, c, h, w, generator=g).cuda()
        # the block's ReLU bitmask as pfst_bn_apply writes it (gamma = 1, beta = 0 on a standardised tensor: bit = out > 0)
        zero, one = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
        _, mask = ops.bn_apply(out, zero, one, one, zero, True, torch.zeros_like(out), want_mask=True)
        side = {}
        for k, x in (('a', xa), ('b', xb)):
            mean = x.mean(dim=(0, 2, 3)).contiguous()
            invstd = torch.rsqrt(x.var(dim=(0, 2, 3), unbiased=False) + 1e-5).contiguous()
            gamma = (torch.rand(c, generator=g) + 0.5).cuda() * (1 if k == 'a' else -1)
            side[k] = dict(x=x, mean=mean, invstd=invstd, gamma=gamma)
        ref = {}
        for k in 'ab':
            sd = side[k]
            dg, db, am = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda'), ops.amax_slots(dy.device)
            dx = ops.bn_backward(dy, None, sd['x'], sd['mean'], sd['invstd'], sd['gamma'], dg, db, True, mask=mask, amax=am)
            ref[k] = (dx, dg, db, float(am.max()))
        got = {k: dict(side[k], dgamma=torch.zeros(c, device='cuda'), dbeta=torch.zeros(c, device='cuda'), amax=ops.amax_slots(dy.device)) for k in 'ab'}
        both = ops.bn_backward_dual(dy, mask, got['a'], got['b'])
        assert both is not None
        torch.cuda.synchronize()
        for k, dx in zip('ab', both):
            rdx, rdg, rdb, ram = ref[k]
            assert float((dx - rdx).abs().max()) <= 1e-6 * float(rdx.abs().max()), k
            assert torch.allclose(got[k]['dgamma'], rdg, rtol=1e-6, atol=1e-6) and torch.allclose(got[k]['dbeta'], rdb, rtol=1e-6, atol=1e-6), k
            assert float(got[k]['amax'].max()) == float(dx.abs().max()), k              # the published maximum is the written tensor's
            assert abs(float(got[k]['amax'].max()) - ram) <= 1e-6 * ram, k
        # against torch autograd of the two-branch sum (fp64)
        xa64, xb64 = xa.double().requires_grad_(True), xb.double().requires_grad_(True)
        ya = torch.nn.functional.batch_norm(xa64, None, None, side['a']['gamma'].double(), None, True, 0.0, 1e-5)
        yb = torch.nn.functional.batch_norm(xb64, None, None, side['b']['gamma'].double(), None, True, 0.0, 1e-5)
        ((ya + yb) * (dy.double() * (out > 0))).sum().backward()
        for dx, x64 in zip(both, (xa64, xb64)):
            assert float((dx.double() - x64.grad).abs().max()) <= 2e-5 * float(x64.grad.abs().max())


def test_first_block_bn_backward_runs_both_layers_in_one_pass():
    """layers.FUSE_BN_BWD_DUAL: one segmentor forward + backward with the dual BatchNorm backward on and off -- four pfst_bn_backward_dual
    launches replace eight pfst_bn_backward ones (layer1-4's first blocks), identical logits, gradients equal to the atomics' order."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd import hip_ops as ops
    from pfst_amd import layers
    from pfst_amd.engine import ParamArena, Tape
    from pfst_amd.registry import build_segmentor
    from pfst_amd.synthetic import synth_batch

    if not layers.FUSE_RES_GATE:
        pytest.skip('needs the gated identity gradient')
    C, b, S = 6, 2, 128
    _, student, _ = seeded_pfgst_state(O, 9)
    batch = synth_batch(b, S, C, seed=43)
    runs, seen_by = {}, {}
    prev = layers.FUSE_BN_BWD_DUAL
    inner = ops.call
    try:
        for dual in (True, False):
            layers.FUSE_BN_BWD_DUAL = dual
            model = build_segmentor(model_cfg(C, 3, dropout=0.0))
            model.load_state_dict(student, strict=True)
            model.cuda()
            arena = ParamArena(list(model.named_parameters()), torch.device('cuda'), with_grad=True)
            model.repack_weights(need_dgrad=True)
            seen = {}

            def counting(name, *a):
                seen[name] = seen.get(name, 0) + 1
                return inner(name, *a)
            ops.call = counting
            try:
                tape = Tape()
                out = model.forward_train(batch['img'].cuda(), batch['img_metas'], ops.to_u8(batch['gt_semantic_seg'].cuda()), None,
                                          return_logits=True, tape=tape)
                tape.backward()
                torch.cuda.synchronize()
            finally:
                ops.call = inner
            runs[dual] = (out['logits'].data.clone(), arena.grad.clone())
            seen_by[dual] = seen
    finally:
        layers.FUSE_BN_BWD_DUAL = prev
    on, off = seen_by[True], seen_by[False]
    assert on.get('pfst_bn_backward_dual', 0) == 4 and off.get('pfst_bn_backward_dual', 0) == 0, (on, off)
    assert off['pfst_bn_backward'] - on['pfst_bn_backward'] == 8, (off['pfst_bn_backward'], on['pfst_bn_backward'])
    assert torch.equal(runs[True][0], runs[False][0])
    _, e = mixed_err(runs[True][1], runs[False][1])
    print(f'   gradient arena, dual BatchNorm backward on vs off: {e:.2e}')
    assert e < 1e-4, e


def test_published_maxima_cover_every_f16x3_operand():
    """f16x3 mode: the scale of an activation operand comes from the slot group its producer(s) published max |.| into -- the normalisation
    pass, the max-pool (layer1's input) and, for the ASPP head's concat, all five writers of the buffer into ONE group (four normalisation
    passes + the image-pool broadcast).  One segmentor forward + backward: every group handed to a GEMM holds exactly the maximum of the
    tensor it describes, and no separate pfst_absmax pass runs over the concat or the pooled map."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd import hip_ops as ops
    from pfst_amd import layers
    from pfst_amd.engine import ParamArena, Tape
    from pfst_amd.registry import build_segmentor
    from pfst_amd.synthetic import synth_batch

    if layers.CONV_MATH != 'f16x3':
        pytest.skip('maxima are only published in f16x3 mode')
    C, b, S = 6, 2, 128
    _, student, _ = seeded_pfgst_state(O, 9)
    batch = synth_batch(b, S, C, seed=5)
    model = build_segmentor(model_cfg(C, 3, dropout=0.1))
    model.load_state_dict(student, strict=True)
    model.cuda()
    ParamArena(list(model.named_parameters()), torch.device('cuda'), with_grad=True)
    model.repack_weights(need_dgrad=True)
    cat_ch = model.decode_head.channels * (len(model.decode_head.dilations) + 1)
    seen, scanned = {}, []
    orig_of, orig_absmax = layers.amax_of, ops.absmax

    predicted = []

    def checked(v):
        had = v.amax is not None
        slots = orig_of(v)
        if had and id(v) not in seen:
            data = v.data
            if v.lazy is not None:
                # a tensor that is never written (conv1 -> bn1 -> relu of a Winograd bottleneck): its group holds the PREDICTED maximum
                # (bn_finalize_partials from the GEMM's min / max partials); the true one from the normalisation pass run here for the test
                data = _data(v)
                predicted.append(tuple(data.shape))
            seen[id(v)] = (tuple(data.shape), slots.max().item(), data.abs().max().item())
        return slots

    def counted(x, *a, **k):
        scanned.append(tuple(x.shape))
        return orig_absmax(x, *a, **k)

    layers.amax_of, ops.absmax = checked, counted
    try:
        tape = Tape()
        model.forward_train(batch['img'].cuda(), batch['img_metas'], ops.to_u8(batch['gt_semantic_seg'].cuda()), None, tape=tape)
        n_fwd_scans = len(scanned)
        tape.backward()
        torch.cuda.synchronize()
    finally:
        layers.amax_of, ops.absmax = orig_of, orig_absmax
    assert len(seen) > 20, len(seen)
    bad = [s for s in seen.values() if s[1] != s[2]]
    assert not bad, bad
    if layers.FOLD_BN_WINO and layers.DEFER_BN_APPLY and layers.WINOGRAD:
        # bn1 of layer2.1-3, layer3.0-5, layer4.0-2 (+ the ASPP concat with its four deferred writers): predicted == true maximum, bit for bit
        assert len(predicted) == 12 + (16 if layers.FOLD_BN_GEMM else 0) + (5 if layers.FOLD_BN_DWSEP else 0) + (1 if layers.FOLD_BN_CONCAT else 0), predicted
    shapes = [s[0] for s in seen.values()]
    assert (b, cat_ch, S // 8, S // 8) in shapes, 'the ASPP concat must arrive with its shared group'
    assert (b, 64, S // 4, S // 4) in shapes or (b, 128, S // 4, S // 4) in shapes, 'the pooled map must arrive with its group'
    assert not [s for s in scanned[:n_fwd_scans] if len(s) == 4 and s[1] in (cat_ch,)], scanned[:n_fwd_scans]


def test_residual_gate_in_the_dgrad_epilogue():
    """layers.FUSE_RES_GATE: dL/d(block input) of a residual block = conv1's data gradient + dL/d(block output) gated by the block's final
    ReLU.  The product forms the sum in conv1's f16x3 data-gradient epilogue (pfst_conv_igemm_f16x3 gate_dy; a downsample layer's
    BatchNorm backward takes the pair as its gated dy) instead of letting the bn3 layer's BatchNorm backward write the gated tensor.  Same
    additions per element: one segmentor forward + backward with the fold on and off gives the same gradients to the summation order of the
    weight gradients' atomics, and the gated tensor is really not written (no dres argument on any residual layer's pfst_bn_backward)."""
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd import hip_ops as ops
    from pfst_amd import layers
    from pfst_amd.engine import ParamArena, Tape
    from pfst_amd.registry import build_segmentor
    from pfst_amd.synthetic import synth_batch

    if layers.CONV_MATH != 'f16x3':
        pytest.skip('the gated epilogue exists on the f16x3 data-gradient kernel')
    C, b, S = 6, 2, 256              # 1/8 maps of 32 x 32 = 1024 pixels: whole 256-element mask groups in layers 2-4; layer1: 64 x 64
    _, student, _ = seeded_pfgst_state(O, 9)
    batch = synth_batch(b, S, C, seed=17)
    runs, n_dres, n_gated = {}, {}, {}
    prev = layers.FUSE_RES_GATE
    try:
        for fold in (True, False):
            layers.FUSE_RES_GATE = fold
            model = build_segmentor(model_cfg(C, 3, dropout=0.0))
            model.load_state_dict(student, strict=True)
            model.cuda()
            arena = ParamArena(list(model.named_parameters()), torch.device('cuda'), with_grad=True)
            model.repack_weights(need_dgrad=True)
            cnt = dict(dres=0, gated=0, flushed=0)
            o_bnb, o_dg, o_rg = ops.bn_backward, ops.conv_dgrad_f16x3, ops.relu_gate_

            def bnb(*a, **k):
                dres = a[9] if len(a) > 9 else k.get('dres')
                cnt['dres'] += dres is not None
                return o_bnb(*a, **k)

            def dg(*a, **k):
                cnt['gated'] += k.get('gate') is not None
                return o_dg(*a, **k)

            def rg(*a, **k):
                cnt['flushed'] += 1
                return o_rg(*a, **k)

            ops.bn_backward, ops.conv_dgrad_f16x3, ops.relu_gate_ = bnb, dg, rg
            try:
                tape = Tape()
                out = model.forward_train(batch['img'].cuda(), batch['img_metas'], ops.to_u8(batch['gt_semantic_seg'].cuda()), None,
                                          return_logits=True, tape=tape)
                tape.backward()
                torch.cuda.synchronize()
            finally:
                ops.bn_backward, ops.conv_dgrad_f16x3, ops.relu_gate_ = o_bnb, o_dg, o_rg
            runs[fold] = (out['logits'].data.clone(), arena.grad.clone())
            n_dres[fold], n_gated[fold] = cnt['dres'], cnt['gated']
            assert cnt['flushed'] == 0, 'every deferred identity gradient must be taken over by a fused consumer'
    finally:
        layers.FUSE_RES_GATE = prev
    # ResNet-50: 16 residual blocks; 4 of them hand the pair to their downsample layer's BatchNorm backward, the other 12 to conv1's epilogue
    assert n_dres == {True: 0, False: 16} and n_gated == {True: 12, False: 0}, (n_dres, n_gated)
    assert torch.equal(runs[True][0], runs[False][0])
    _, e = mixed_err(runs[True][1], runs[False][1])
    assert e < 1e-5, e
