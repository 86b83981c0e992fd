"""Fused BatchNorm-backward sums (csrc/conv_epilogue.h, pfst_bnb_fuse_t): the data-gradient launch that completes dL/dy of a
conv -> BN -> [+res] -> ReLU layer also emits sum dz and sum dz*x from its epilogue, and pfst_bn_backward skips its reduction
pass.  Checked against the two-pass kernels (same inputs, same gate) and against torch autograd in fp64."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def ops():
    from pfst_amd import hip_ops
    return hip_ops


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# (n, channels of the BN layer = Cin of the consumer conv, consumer Cout, H, W, ksize, dil, mode, accumulate)
CASES = [
    (2, 128, 256, 16, 16, 1, 1, 'gate_x', False),     # bn2 -> conv3 (1x1), ReLU gate recomputed from x
    (2, 256, 64, 16, 24, 1, 1, 'gate_y', True),       # block output (residual) <- next block's conv1, accumulate over the residual path
    (2, 64, 64, 12, 16, 3, 1, 'gate_x', False),       # BM = 64 tile, 3x3 consumer (layer1 conv2)
    (1, 32, 32, 20, 20, 3, 1, 'gate_x', True),        # BM = 32 tile (stem), ragged pixel tile (400 px), accumulate
    (2, 128, 128, 9, 15, 3, 2, 'none', False),        # no ReLU, dilated consumer, ragged pixels
    (3, 512, 128, 8, 16, 1, 1, 'gate_y', False),      # 4 row tiles
]


@pytest.mark.parametrize('math', ['f32', 'bf16x6', 'f16x3'])
@pytest.mark.parametrize('n,c,co,h,w,k,dil,mode,acc', CASES)
def test_fused_bn_backward_sums_match_two_pass_and_autograd(ops, n, c, co, h, w, k, dil, mode, acc, math):
    g = torch.Generator().manual_seed(c + co + h)
    pad = dil if k == 3 else 0
    pre = (torch.randn(n, c, h, w, generator=g) * 1.7 + 0.8)              # pre-BN tensor of the owner layer (non-zero mean)
    gamma = 0.6 + 0.8 * torch.rand(c, generator=g)
    beta = 0.3 * torch.randn(c, generator=g)
    res = torch.randn(n, c, h, w, generator=g) if mode == 'gate_y' else None
    relu = mode != 'none'
    wc = torch.randn(co, c, k, k, generator=g) * (2.0 / (c * k * k)) ** 0.5   # consumer conv
    dyc = torch.randn(n, co, h, w, generator=g)                            # gradient wrt the consumer conv's output
    old = torch.randn(n, c, h, w, generator=g) if acc else None            # what earlier writers left in the gradient buffer

    # ---- fp64 autograd reference of the owner layer's backward, driven by the consumer's data gradient (+ old)
    x64 = pre.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    z = F.batch_norm(x64, None, None, g64, b64, True, 0.0, 1e-5)
    if res is not None:
        z = z + res.double()
    y64 = F.relu(z) if relu else z
    yin = y64.detach().requires_grad_(True)
    F.conv2d(yin, wc.double(), None, 1, pad, dil).backward(dyc.double())
    dy_total = yin.grad + (old.double() if acc else 0.0)
    dx_ref, dg_ref, db_ref = torch.autograd.grad(y64, [x64, g64, b64], dy_total)

    # ---- HIP: forward statistics (+ coef), apply, then the consumer's data gradient with / without the fused sums
    xd = pre.to(DEV)
    mean, invstd, coef = ops.bn_stats(xd, gamma=gamma.to(DEV), beta=beta.to(DEV))
    gd, bd = gamma.to(DEV), beta.to(DEV)
    y = ops.bn_apply(xd, mean, invstd, gd, bd, relu, None if res is None else res.to(DEV))
    assert rel(y, y64) < 1e-5
    if math == 'f32':
        _, wd = ops.pack_weight(wc.to(DEV))
        dgrad = ops.conv_dgrad
    elif math == 'bf16x6':                  # the fp32-faithful bf16x6 data-gradient kernel carries the same fused epilogue
        _, wd = ops.pack_weight_split(wc.to(DEV))
        dgrad = ops.conv_dgrad_split
    else:                                   # ... and so does the f16x3 one (whole 128-row tiles, contraction over whole 32-channel blocks)
        if not (ops.f16x3_eligible(co, c) and c % 128 == 0):
            pytest.skip('shape outside the f16x3 kernel')
        _, wd, wa = ops.pack_weight_f16x2(wc.to(DEV), False, True)

        def dgrad(dy, w4, cin, hw, ks, s, d, p, out, accumulate, bnb):
            return ops.conv_dgrad_f16x3(dy, w4, wa, ops.absmax(dy), cin, hw, ks, s, d, p, out=out, accumulate=accumulate, bnb=bnb)
    outs = {}
    for fused in (False, True):
        buf = old.to(DEV).clone() if acc else torch.empty(n, c, h, w, device=DEV)
        bnb = (xd, y if mode == 'gate_y' else None, coef, relu) if fused else None
        r = dgrad(dyc.to(DEV), wd, c, (h, w), k, 1, dil, pad, out=buf, accumulate=acc, bnb=bnb)
        part, slots = (r[1], r[2]) if fused else (None, 0)
        dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        dx = ops.bn_backward(buf, y if mode == 'gate_y' else None, xd, mean, invstd, gd, dg, db, relu, beta=bd, partials=part, slots=slots)
        outs[fused] = (dx, dg, db, buf)
    assert torch.equal(outs[True][3], outs[False][3]), 'the data gradient itself must not depend on the fusion'
    assert rel(outs[True][3], dy_total) < 1e-5
    for i, (name, ref) in enumerate((('dx', dx_ref), ('dgamma', dg_ref), ('dbeta', db_ref))):
        assert rel(outs[False][i], ref) < 2e-5, (name, 'two-pass', rel(outs[False][i], ref))
        assert rel(outs[True][i], ref) < 2e-5, (name, 'fused', rel(outs[True][i], ref))
        assert rel(outs[True][i], outs[False][i]) < 1e-5, (name, 'fused vs two-pass')


def test_fused_launch_rejects_unsupported_shapes(ops):
    from pfst_amd._lib import PfstHipError
    n, c, co, h = 1, 48, 64, 8            # 48 rows is not a whole 64-row tile
    x = torch.randn(n, c, h, h, device=DEV)
    mean, invstd, coef = ops.bn_stats(x, gamma=torch.ones(c, device=DEV), beta=torch.zeros(c, device=DEV))
    _, wd = ops.pack_weight(torch.randn(co, c, 1, 1, device=DEV))
    with pytest.raises((PfstHipError, AssertionError)):
        ops.conv_dgrad(torch.randn(n, co, h, h, device=DEV), wd, c, (h, h), 1, bnb=(x, None, coef, True))


def test_a_late_gradient_writer_after_the_fused_launch_is_an_error():
    from pfst_amd.engine import Var
    v = Var(torch.zeros(1, 4, 2, 2, device=DEV), True)
    assert v.claim_first_use() and not v.claim_first_use()
    v.grad_target()
    v.grad_target(final=True)
    with pytest.raises(RuntimeError):
        v.grad_target()


@pytest.mark.parametrize('case', [(2, 128, 128, 16, 16, 1, False, True), (2, 128, 256, 16, 24, 2, True, True), (1, 256, 128, 32, 32, 4, False, False)])
def test_wino_output_emits_the_bn_backward_sums(ops, case):
    """pfst_wino_output(bnb_x): the output transform of a Winograd DATA-GRADIENT launch that completes the gradient of a conv -> BN [-> ReLU]
    layer's output (Bottleneck conv1 -> bn1 -> relu behind a Winograd conv2) also emits that layer's BatchNorm-backward partials, and
    pfst_bn_backward skips its reduction pass: same dx / dgamma / dbeta as the two-pass kernels on the same gradient tensor (to the
    summation order), with accumulation into an earlier writer's values and without the ReLU."""
    n, c, co, h, w, dil, acc, relu = case
    g = torch.Generator().manual_seed(c + co + h)
    pre = (torch.randn(n, c, h, w, generator=g) * 1.7 + 0.8).to(DEV)
    gamma = (0.6 + 0.8 * torch.rand(c, generator=g)).to(DEV)
    beta = (0.3 * torch.randn(c, generator=g)).to(DEV)
    wc = (torch.randn(co, c, 3, 3, generator=g) * (2.0 / (c * 9)) ** 0.5).to(DEV)
    dyc = torch.randn(n, co, h, w, generator=g).to(DEV)
    old = torch.randn(n, c, h, w, generator=g).to(DEV) if acc else None
    mean, invstd, coef = ops.bn_stats(pre, gamma=gamma, beta=beta)
    _, ud = ops.wino_pack_weight(wc, False, True)
    plain = ops.wino_conv(dyc, ud, c, dil, out=old.clone() if acc else None, accumulate=acc)
    fused, part, slots = ops.wino_conv(dyc, ud, c, dil, out=old.clone() if acc else None, accumulate=acc, bnb=(pre, coef, relu))
    assert torch.equal(fused, plain)
    outs = []
    for p, s in ((None, 0), (part, slots)):
        dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        dx = ops.bn_backward(plain, None, pre, mean, invstd, gamma, dg, db, relu=relu, beta=beta, partials=p, slots=s)
        outs.append((dx, dg, db))
    for a, b, what in zip(outs[1], outs[0], ('dx', 'dgamma', 'dbeta')):
        assert rel(a, b) < 2e-6, (what, rel(a, b))
