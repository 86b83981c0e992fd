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


@pytest.mark.parametrize('math', ['f32', 'bf16x6'])
@pytest.mark.parametrize('wino', [True, False])
def test_every_backward_link_as_wired(wino, math):
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
    prev, prev_math = layers.WINOGRAD, layers.CONV_MATH
    layers.WINOGRAD, layers.CONV_MATH = wino, math      # both arithmetics of the dense convolutions: fp32-input MFMA and the bf16x6 split
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

        def reference(tag, dy):
            """fp64 CPU autograd of the oracle's restatement of this one link at the HIP layer's actual input"""
            op = tag['op']
            x = tag['x'].data.detach().cpu().double().requires_grad_(tag['x'].requires_grad)
            leaves, keys = [x] if x.requires_grad else [], ['x'] if x.requires_grad else []
            if op in ('conv_bn_act', 'conv'):
                cv = tag['conv']
                w = cv.weight.data.detach().cpu().double().requires_grad_(True)
                leaves.append(w); keys.append('weight')
                bias = None
                if cv.bias is not None:
                    bias = cv.bias.data.detach().cpu().double().requires_grad_(True)
                    leaves.append(bias); keys.append('bias')
                y = F.conv2d(x, w, bias, cv.stride, cv.padding, cv.dilation, cv.groups)
                if op == 'conv_bn_act':
                    gam = tag['bn'].weight.data.detach().cpu().double().requires_grad_(True)
                    bet = tag['bn'].bias.data.detach().cpu().double().requires_grad_(True)
                    leaves += [gam, bet]; keys += ['gamma', 'beta']
                    y = F.batch_norm(y, None, None, gam, bet, True, 0.0, O.BN_EPS)
                    if tag['residual'] is not None:
                        r = tag['residual'].data.detach().cpu().double().requires_grad_(tag['residual'].requires_grad)
                        if r.requires_grad:
                            leaves.append(r); keys.append('residual')
                        y = y + r
                    if tag['relu']:
                        gate = (tag['out'].data.detach().cpu() > 0).double()      # the HIP forward's own ReLU decision
                        fwd = F.relu(y).detach()
                        y = y * gate
                    else:
                        fwd = y.detach()
                    state['fwd_err'] = mixed_err(tag['out'].data, fwd)[1]
            elif op == 'maxpool':
                y = F.max_pool2d(x, 3, 2, 1)
            elif op in ('resize', 'broadcast'):
                y = F.interpolate(x, size=tag['out'].data.shape[-2:], mode='bilinear', align_corners=False)
            elif op == 'gap':
                y = x.mean((2, 3), keepdim=True)
            elif op == 'ce':
                lab = tag['label'].detach().cpu()
                pw = None if tag['weight'] is None else tag['weight'].detach().cpu().double()
                up = F.interpolate(x, size=lab.shape[-2:], mode='bilinear', align_corners=False)
                y = O.ce_loss(up, lab.reshape(lab.shape[0], *lab.shape[-2:]).long(), pw, tag['class_weight'], tag['loss_weight'],
                              tag['ignore_index'])
                dy = torch.ones((), dtype=torch.float64)
            else:
                raise AssertionError(f'untested tape op {op}')
            gs = torch.autograd.grad(y, leaves, dy.double())
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
                             state.get('bn_count')))
            state.clear()

        tape = Tape(observer)
        gt8 = ops.to_u8(gt.cuda())
        model.forward_train(img.cuda(), batch['img_metas'], gt8, pixw.cuda(), tape=tape)
        n_closures = len(tape.fns)
        tape.backward()
        torch.cuda.synchronize()
    finally:
        layers.WINOGRAD, layers.CONV_MATH = prev, prev_math

    print(f'\n{len(rows)} checked tensors over {n_closures} closures (winograd={wino}); worst element error / bound, norm-wise rel:')
    for op, name, k, worst, nrm, fe, de, _ in sorted(rows, key=lambda r: -r[3])[:25]:
        print(f'   {worst:8.3f} {nrm:9.2e}  fwd {fe:8.1e}  chain-dy {de:8.1e}  {name}:{k} [{op}]')
    print(f'   {n_fused[0]} BatchNorm layers had received fused backward sums from the launch completing their gradient')
    if layers.FUSE_BN_BWD and layers.FUSE_BN_BWD_MIN_K <= 512:                     # the defaults (not PFST_FUSE_BN_BWD=0 / a higher threshold)
        assert n_fused[0] >= (20 if math == 'f32' else 0)  # layers.FUSE_BN_BWD_MIN_K = 512: the MFMA-bound fp32 data-gradient launches only
    checked_ops = {r[0] for r in rows}
    assert {'conv_bn_act', 'conv', 'maxpool', 'resize', 'gap', 'broadcast', 'ce'} <= checked_ops, checked_ops
    n_bn = sum(1 for m in model.modules() if isinstance(m, layers.BatchNorm2dP))
    assert n_bn == 70 and sum(1 for r in rows if r[0] == 'conv_bn_act' and r[2] == 'weight') == n_bn      # every conv+BN layer
    # every link: element-wise mixed bound.  One exception, stated: the image-pool branch normalises b = 2 values per channel, its
    # backward is a cancellation conditioned by 1/var of two numbers (fp32 forward of that layer: 2e-4) -- norm-wise 1e-3 there.
    small = [r for r in rows if r[7] is not None and r[7] < 16]
    assert {r[1] for r in small} == {'decode_head.image_pool.1.conv'}
    bad = [r for r in rows if r not in small and not r[3] <= 1.0] + [r for r in small if not r[4] <= 1e-3]
    assert not bad, bad[:10]
