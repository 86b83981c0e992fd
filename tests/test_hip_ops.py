"""Per-kernel parity of libpfst_hip.so (through the C ABI) against plain PyTorch CPU fp32/fp64 math.
Tolerance: north_star's 1e-3 relative for floating point (most kernels are far tighter); exact for
integer / index outputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from pfst_amd import hip_ops
    return hip_ops


def g(seed=0):
    return torch.Generator().manual_seed(seed)


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def assert_close(a, b, tol=1e-3, what=''):
    e = rel_err(a, b)
    assert e < tol, f'{what} rel err {e:.3e} >= {tol}'


CONV_CASES = [
    # n, cin, cout, H, W, k, stride, dil, pad
    (2, 3, 32, 33, 37, 3, 2, 1, 1),      # stem.0 (generic K path, stride 2, odd sizes)
    (2, 32, 64, 20, 24, 3, 1, 1, 1),     # stem.6
    (2, 64, 256, 16, 16, 1, 1, 1, 0),    # 1x1
    (1, 128, 128, 17, 19, 3, 2, 1, 1),   # layer2.0.conv2 (stride 2)
    (2, 256, 512, 12, 12, 1, 2, 1, 0),   # layer2.0.downsample (1x1 stride 2)
    (2, 64, 64, 14, 18, 3, 1, 2, 2),     # dilated d=2
    (1, 48, 80, 15, 15, 3, 1, 4, 4),     # dilated d=4, Cout not multiple of 32
    (2, 512, 6, 16, 16, 1, 1, 1, 0),     # conv_seg (tiny M, bias)
    (2, 10, 32, 16, 16, 3, 2, 1, 1),     # 10-band stem
    (1, 160, 48, 9, 130, 1, 1, 1, 0),    # c1_bottleneck-like, P not multiple of 128
    # shapes on the K-quad weight-gradient path (stride 1, width % 4 == 0): border quads, minimum width, ragged J / M
    (2, 24, 40, 12, 16, 3, 1, 2, 2),
    (1, 16, 96, 8, 4, 3, 1, 1, 1),
    (2, 8, 16, 10, 12, 3, 1, 4, 4),
    (1, 72, 200, 6, 20, 1, 1, 1, 0),
    (2, 32, 64, 40, 36, 3, 1, 1, 1),     # > 512 pixels: split-K chunks
    (2, 16, 48, 12, 64, 3, 1, 1, 1),     # 3x3 row-walk path (width % 16 == 0): interior + edge steps, dil 1 / 2 / 4
    (1, 40, 72, 20, 48, 3, 1, 2, 2),
    (2, 24, 136, 18, 80, 3, 1, 4, 4),
    (1, 16, 32, 9, 16, 3, 1, 1, 1),      # one K-step per row: every step is an edge step
    (2, 64, 256, 32, 32, 1, 1, 1, 0),    # 8 pixel tiles: the XCD-aware tile order + the buffer-store epilogue on full tiles
    (1, 32, 160, 32, 64, 3, 1, 2, 2),    # 16 pixel tiles, M = 128 + 32: fast and generic epilogue paths in one launch
    # contractions >= 512 deep with more than 64 output rows: the software-pipelined bf16x6 loops (conv_split.hip) -- ragged pixel tile,
    # ragged row tile + tap switches between the branch-free blocks, 6 K-steps per tap
    (2, 512, 256, 16, 24, 1, 1, 1, 0),
    (2, 512, 256, 15, 20, 1, 1, 1, 0),   # 300 pixels: ragged last pixel tile on the K = 32 pairing
    (1, 560, 144, 12, 12, 1, 1, 1, 0),   # 560 channels (16 but not 32 per step): the K = 16 pipelined loop, ragged rows and pixels
    (1, 64, 192, 20, 28, 3, 1, 2, 2),
    (2, 96, 128, 12, 20, 3, 1, 1, 1),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fprop_dgrad_wgrad(ops, case):
    n, ci, co, H, W, k, s, d, p = case
    x = torch.randn(n, ci, H, W, generator=g(1))
    w = torch.randn(co, ci, k, k, generator=g(2)) * 0.1
    b = torch.randn(co, generator=g(3)) if co == 6 else None
    xr = x.clone().requires_grad_()
    wr = w.clone().requires_grad_()
    y_ref = F.conv2d(xr, wr, b, s, p, d)
    dy = torch.randn(y_ref.shape, generator=g(4))
    y_ref.backward(dy)
    xd, wd_, dyd = x.to(DEV), w.to(DEV), dy.to(DEV)
    wf, wdg = ops.pack_weight(wd_)
    y = ops.conv_fprop(xd, wf, co, k, s, d, p, bias=None if b is None else b.to(DEV))
    assert_close(y, y_ref, 2e-5, 'fprop')
    dx = ops.conv_dgrad(dyd, wdg, ci, (H, W), k, s, d, p)
    assert_close(dx, xr.grad, 2e-5, 'dgrad')
    # accumulate mode
    dx2 = ops.conv_dgrad(dyd, wdg, ci, (H, W), k, s, d, p, out=dx.clone(), accumulate=True)
    assert_close(dx2, 2 * xr.grad, 2e-5, 'dgrad-acc')
    dw = torch.zeros_like(wd_)
    ops.conv_wgrad_(dw, xd, dyd, k, s, d, p)
    assert_close(dw, wr.grad, 5e-5, 'wgrad')
    if b is not None:
        db = torch.zeros(co, device=DEV)
        ops.bias_grad_(db, dyd)
        assert_close(db, dy.sum((0, 2, 3)), 1e-5, 'bias grad')


@pytest.mark.parametrize('case', [c for c in CONV_CASES if c[1] % 16 == 0])
def test_conv_split_bf16x6_is_fp32_faithful(ops, case):
    """6-term bf16 split on the bf16 matrix cores: as close to fp64 as fp32 arithmetic (1e-6), fprop + dgrad."""
    n, ci, co, H, W, k, s, d, p = case
    x = torch.randn(n, ci, H, W, generator=g(1))
    w = torch.randn(co, ci, k, k, generator=g(2)) * 0.1
    ref = F.conv2d(x.double(), w.double(), None, s, p, d)
    w6f, w6d = ops.pack_weight_split(w.to(DEV), True, co % 16 == 0)
    y = ops.conv_fprop_split(x.to(DEV), w6f, co, k, s, d, p)
    assert_close(y, ref, 2e-6, 'split fprop')
    if co % 16 == 0:
        dy = torch.randn(ref.shape, generator=g(4))
        dx_ref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), s, p, d)
        dx = ops.conv_dgrad_split(dy.to(DEV), w6d, ci, (H, W), k, s, d, p)
        assert_close(dx, dx_ref, 2e-6, 'split dgrad')
        dx2 = ops.conv_dgrad_split(dy.to(DEV), w6d, ci, (H, W), k, s, d, p, out=dx.clone(), accumulate=True)
        assert_close(dx2, 2 * dx_ref, 2e-6, 'split dgrad-acc')


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_wgrad_split(ops, case):
    n, ci, co, H, W, k, s, d, p = case
    x = torch.randn(n, ci, H, W, generator=g(1))
    ho, wo = (H + 2 * p - (k - 1) * d - 1) // s + 1, (W + 2 * p - (k - 1) * d - 1) // s + 1
    dy = torch.randn(n, co, ho, wo, generator=g(4))
    ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, k, k), dy.double(), s, p, d)
    dw = torch.zeros(co, ci, k, k, device=DEV)
    ops.conv_wgrad_split_(dw, x.to(DEV), dy.to(DEV), k, s, d, p)
    assert_close(dw, ref, 3e-6, 'split wgrad')


@pytest.mark.parametrize('split', [False, True])
@pytest.mark.parametrize('case', [c for c in CONV_CASES if c[2] != 6])
def test_conv_fused_bn_statistics(ops, case, split):
    """BN batch statistics from the conv epilogue's partial sums == statistics of the conv output (torch, fp64),
    including the running-stat update (momentum 0.1, unbiased variance)."""
    n, ci, co, H, W, k, s, d, p = case
    if split and ci % 16:
        pytest.skip('split kernel needs Cin % 16 == 0')
    x = torch.randn(n, ci, H, W, generator=g(1)) + 0.3
    w = torch.randn(co, ci, k, k, generator=g(2)) * 0.1
    ref = F.conv2d(x.double(), w.double(), None, s, p, d)
    if split:
        w6f, _ = ops.pack_weight_split(w.to(DEV), True, False)
        y, st, slots = ops.conv_fprop_split(x.to(DEV), w6f, co, k, s, d, p, want_stats=True)
    else:
        wf, _ = ops.pack_weight(w.to(DEV), want_dgrad=False)
        y, st, slots = ops.conv_fprop(x.to(DEV), wf, co, k, s, d, p, want_stats=True)
    assert_close(y, ref, 2e-5, 'fprop with stats')
    rm, rv = torch.zeros(co, device=DEV), torch.ones(co, device=DEV)
    cnt = ref.numel() // co
    mean, invstd = ops.bn_finalize_partials(st, slots, co, cnt, rm, rv, 0.1, 1e-5)
    m_ref = ref.mean((0, 2, 3))
    v_ref = ref.var((0, 2, 3), unbiased=False)
    assert_close(mean, m_ref, 2e-5, 'mean')
    assert_close(invstd, 1.0 / torch.sqrt(v_ref + 1e-5), 2e-5, 'invstd')
    assert_close(rm, 0.1 * m_ref, 2e-5, 'running mean')
    assert_close(rv, 0.9 + 0.1 * v_ref * cnt / (cnt - 1), 2e-5, 'running var')
    # and identical (to fp32 rounding of the partials) to the stand-alone statistics kernel
    m2, i2 = ops.bn_stats(y)
    assert_close(mean, m2, 1e-5, 'mean vs bn_stats')
    assert_close(invstd, i2, 1e-5, 'invstd vs bn_stats')


WINO_CASES = [
    # n, cin, cout, H, W, dil
    (2, 32, 48, 16, 16, 1),
    (2, 16, 32, 12, 20, 2),
    (1, 64, 32, 16, 24, 4),
    (2, 16, 16, 9, 13, 1),        # odd sizes: ragged last tile
    (1, 32, 16, 10, 14, 2),       # sub-grids of different sizes
    (2, 48, 80, 8, 8, 1),         # Cout not a multiple of 32
]


# fp32 Winograd against the fp64 direct convolution, relative to max |ref| (rel_err): F(2x2) is as good as the direct fp32
# kernel; F(4x4) has transform entries up to 8 and 1/24, so its rounding error is ~10x larger -- still 30x inside the 1e-3 bar
WINO_TOL = {2: 3e-6, 4: 3e-5}


@pytest.mark.parametrize('m', [2, 4])
@pytest.mark.parametrize('case', WINO_CASES)
def test_winograd_matches_direct_convolution(ops, case, m):
    """Winograd F(m x m,3x3) forward, data gradient (incl. accumulate) and weight gradient == F.conv2d / autograd (fp64 ref)."""
    n, ci, co, H, W, d = case
    tol = WINO_TOL[m]
    x = torch.randn(n, ci, H, W, generator=g(1))
    w = torch.randn(co, ci, 3, 3, generator=g(2)) * 0.1
    dy = torch.randn(n, co, H, W, generator=g(4))
    ref = F.conv2d(x.double(), w.double(), None, 1, d, d)
    dx_ref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), 1, d, d)
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), 1, d, d)
    uf, ud = ops.wino_pack_weight(w.to(DEV), m=m)
    y = ops.wino_conv(x.to(DEV), uf, co, d, m=m)
    assert_close(y, ref, tol, 'winograd fprop')
    # fused BatchNorm statistics of the output transform
    y2, st, slots = ops.wino_conv(x.to(DEV), uf, co, d, want_stats=True, m=m)
    mean, invstd = ops.bn_finalize_partials(st, slots, co, n * H * W)
    assert_close(y2, ref, tol, 'winograd fprop with stats')
    assert_close(mean, ref.mean((0, 2, 3)), 2e-5, 'winograd stats mean')
    assert_close(invstd, 1.0 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5), 2e-5, 'winograd stats invstd')
    dx = ops.wino_conv(dy.to(DEV), ud, ci, d, m=m)
    assert_close(dx, dx_ref, tol, 'winograd dgrad')
    dx2 = ops.wino_conv(dy.to(DEV), ud, ci, d, out=dx.clone(), accumulate=True, m=m)
    assert_close(dx2, 2 * dx_ref, tol, 'winograd dgrad accumulate')
    if ops.wino_tiles(H, W, d, m) % 4 == 0:
        dw = torch.zeros(co, ci, 3, 3, device=DEV)
        ops.wino_wgrad_(dw, x.to(DEV), dy.to(DEV), d, m=m)
        assert_close(dw, dw_ref, 2 * tol, 'winograd wgrad')
        # with the transformed input kept from the forward pass, as the train step does
        _, (v, v_amax) = ops.wino_conv(x.to(DEV), uf, co, d, keep_v=True, m=m)      # (V, its amax slot group: None outside f16x3)
        assert v_amax is None
        dw2 = torch.zeros(co, ci, 3, 3, device=DEV)
        ops.wino_wgrad_(dw2, x.to(DEV), dy.to(DEV), d, v=v, m=m)
        assert_close(dw2, dw_ref, 2 * tol, 'winograd wgrad from the kept V')


def test_winograd_tile_counts(ops):
    """tiles per image: d*d sub-grids of ceil(Hs/m) x ceil(Ws/m) tiles"""
    assert ops.wino_tiles(128, 128, 1, 2) == 64 * 64 and ops.wino_tiles(128, 128, 1, 4) == 32 * 32
    assert ops.wino_tiles(128, 128, 4, 4) == 16 * 8 * 8 and ops.wino_tiles(9, 13, 1, 4) == 3 * 4
    assert ops.wino_tiles(10, 14, 2, 2) == 4 * 3 * 4 and ops.wino_tiles(10, 14, 2, 4) == 4 * 2 * 2


@pytest.mark.parametrize('m', [2, 4])
@pytest.mark.parametrize('case', [c for c in WINO_CASES if c[1] % 16 == 0 and c[2] % 16 == 0])
def test_winograd_on_the_bf16x6_gemm(ops, case, m):
    """The same Winograd pipeline with the transform-domain GEMMs on the fp32-faithful bf16x6 kernel."""
    n, ci, co, H, W, d = case
    x = torch.randn(n, ci, H, W, generator=g(1))
    w = torch.randn(co, ci, 3, 3, generator=g(2)) * 0.1
    dy = torch.randn(n, co, H, W, generator=g(4))
    ref = F.conv2d(x.double(), w.double(), None, 1, d, d)
    dx_ref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), 1, d, d)
    uf, ud = ops.wino_pack_weight_split(w.to(DEV), m=m)
    assert uf.dtype == torch.uint8
    assert_close(ops.wino_conv(x.to(DEV), uf, co, d, m=m), ref, WINO_TOL[m], 'winograd/bf16x6 fprop')
    assert_close(ops.wino_conv(dy.to(DEV), ud, ci, d, m=m), dx_ref, WINO_TOL[m], 'winograd/bf16x6 dgrad')


F16_CASES = [c for c in CONV_CASES if c[1] % 32 == 0 and c[2] > 32] + [
    # 33 ... 64 output rows: the 64-row tile (layer1 conv2, its data gradient, stem.6; a ragged 48-row case, stride 2, a 1x1)
    (2, 64, 64, 32, 32, 3, 1, 1, 1), (1, 32, 48, 20, 24, 3, 1, 1, 1), (2, 64, 64, 17, 19, 3, 2, 1, 1), (2, 256, 64, 16, 16, 1, 1, 1, 0),
    # 1x1 contractions that end in half a channel block (the decoder's 560 -> 512 pointwise convolution; 48: a single, half-empty pair)
    (2, 560, 512, 16, 24, 1, 1, 1, 0), (1, 48, 96, 12, 12, 1, 1, 1, 0), (2, 80, 160, 9, 14, 1, 2, 1, 0)]


@pytest.mark.parametrize('spread', [0.0, 2.5])
@pytest.mark.parametrize('case', F16_CASES)
def test_conv_f16x3_is_fp32_faithful(ops, case, spread):
    """Two scaled fp16 pieces per operand, three fp16 MFMAs per product (csrc/conv_f16x3.hip): as close to fp64 as fp32 arithmetic, fprop
    + dgrad + the 1x1 weight gradient.  spread > 0: log-normal magnitudes over ~6 decades and tensors far from 1 (1e-6 gradients, 3e4
    activations): what the per-tensor power-of-two scales are for -- without them fp16 would flush or overflow."""
    n, ci, co, H, W, k, s, d, p = case
    x = torch.randn(n, ci, H, W, generator=g(1))
    w = torch.randn(co, ci, k, k, generator=g(2)) * 0.1
    if spread:
        x = x * 3e4 * torch.exp(spread * torch.randn(x.shape, generator=g(7)))
        w = w * 1e-3 * torch.exp(0.5 * spread * torch.randn(w.shape, generator=g(8)))
    ref = F.conv2d(x.double(), w.double(), None, s, p, d)
    xd, wd = x.to(DEV), w.to(DEV)
    w4f, w4d, wa = ops.pack_weight_f16x2(wd, True, ops.f16x3_eligible(co, ci, k))
    xa = ops.absmax(xd)
    assert float(xa.max()) == float(x.abs().max()) and float(wa.max()) == float(w.abs().max())       # the slot group holds the exact maximum
    y = ops.conv_fprop_f16x3(xd, w4f, wa, xa, co, k, s, d, p)
    assert_close(y, ref, 2e-6, 'f16x3 fprop')
    # BatchNorm statistics from the epilogue's partials (every tile height writes two slots per 128-pixel tile): those of the output itself
    y_st, st, slots = ops.conv_fprop_f16x3(xd, w4f, wa, xa, co, k, s, d, p, want_stats=True)
    assert torch.equal(y_st, y)
    mean, invstd = ops.bn_finalize_partials(st, slots, co, ref.numel() // co)
    m2, i2 = ops.bn_stats(y)
    assert_close(mean, m2, 1e-5, 'f16x3 epilogue mean vs bn_stats')
    assert_close(invstd, i2, 1e-5, 'f16x3 epilogue invstd vs bn_stats')
    # the bf16x6 kernel on the same data, for scale: f16x3 is not allowed to be worse than 1.5x + 1e-7
    w6f, _ = ops.pack_weight_split(wd, True, False)
    e6 = rel_err(ops.conv_fprop_split(xd, w6f, co, k, s, d, p), ref)
    assert rel_err(y, ref) <= 1.5 * e6 + 1e-7
    dy = torch.randn(ref.shape, generator=g(4))
    if spread:
        dy = dy * 1e-6 * torch.exp(spread * torch.randn(dy.shape, generator=g(9)))
    if w4d is not None:
        dx_ref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), s, p, d)
        da = ops.absmax(dy.to(DEV))
        dx = ops.conv_dgrad_f16x3(dy.to(DEV), w4d, wa, da, ci, (H, W), k, s, d, p)
        assert_close(dx, dx_ref, 2e-6, 'f16x3 dgrad')
        dx2 = ops.conv_dgrad_f16x3(dy.to(DEV), w4d, wa, da, ci, (H, W), k, s, d, p, out=dx.clone(), accumulate=True)
        assert_close(dx2, 2 * dx_ref, 2e-6, 'f16x3 dgrad-acc')
    if k == 1 and s == 1 and (H * W) % 4 == 0 and co > 64:          # (the f16x3 weight gradient has no 64-row tile)
        dw_ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, 1, 1), dy.double(), 1, 0, 1)
        dw = torch.zeros(co, ci, 1, 1, device=DEV)
        ops.conv_wgrad_f16x3_(dw, xd, dy.to(DEV), xa, ops.absmax(dy.to(DEV)))
        assert_close(dw, dw_ref, 3e-6, 'f16x3 wgrad')
        ops.conv_wgrad_f16x3_(dw, xd, dy.to(DEV), xa, ops.absmax(dy.to(DEV)))           # accumulates (fp32 atomics)
        assert_close(dw, 2 * dw_ref, 3e-6, 'f16x3 wgrad accumulate')


@pytest.mark.parametrize('case', [(2, 256, 128, 16, 16, 1, 1, 1, 0), (2, 128, 160, 16, 32, 3, 1, 2, 2), (1, 512, 128, 32, 32, 1, 1, 1, 0)])
def test_f16x3_dgrad_adds_the_gated_identity_gradient(ops, case):
    """pfst_conv_igemm_f16x3(gate_dy, gate_mask): out = data gradient + (bit ? g : 0) -- the identity branch of a residual block folded into
    conv1's epilogue.  Exactly the sum of the plain launch and the gated tensor written by itself (pfst_relu_gate), with the bitmask
    pfst_bn_apply writes; also with the fused BatchNorm-backward sums in the same epilogue (they must see the SUM)."""
    n, ci, co, H, W, k, s, d, p = case
    x = torch.randn(n, ci, H, W, generator=g(1))
    w = (torch.randn(co, ci, k, k, generator=g(2)) * 0.1).to(DEV)
    dy = torch.randn(n, co, H, W, generator=g(4)).to(DEV)
    _, w4d, wa = ops.pack_weight_f16x2(w, False, True)
    da = ops.absmax(dy)
    assert ops.dgrad_gate_ok(ci, (H, W))
    gsrc = torch.randn(x.shape, generator=g(11)).to(DEV)
    pre = torch.randn(x.shape, generator=g(12)).to(DEV)
    one, zero = torch.ones(ci, device=DEV), torch.zeros(ci, device=DEV)
    mean, invstd, coef = ops.bn_stats(pre, gamma=one, beta=zero)
    yb, mask = ops.bn_apply(pre, mean, invstd, one, zero, True, residual=torch.zeros_like(pre), want_mask=True)
    assert mask is not None
    gated = ops.relu_gate_(torch.empty_like(gsrc), gsrc, mask)
    assert torch.equal(gated, torch.where(yb > 0, gsrc, torch.zeros_like(gsrc)))
    dx = ops.conv_dgrad_f16x3(dy, w4d, wa, da, ci, (H, W), k, s, d, p)
    dxg = ops.conv_dgrad_f16x3(dy, w4d, wa, da, ci, (H, W), k, s, d, p, gate=(gsrc, mask))
    assert torch.equal(dxg, dx + gated)
    assert torch.equal(ops.relu_gate_(dx.clone(), gsrc, mask, accumulate=True), dx + gated)
    # with the BatchNorm-backward sums of the layer that owns the gradient (a residual layer: gate from y): sums of the gated total
    pre2 = torch.randn(x.shape, generator=g(13)).to(DEV)
    m2, i2, coef2 = ops.bn_stats(pre2, gamma=one, beta=zero)
    y2 = ops.bn_apply(pre2, m2, i2, one, zero, True, residual=torch.randn(x.shape, generator=g(14)).to(DEV))
    out_a, part_a, slots = ops.conv_dgrad_f16x3(dy, w4d, wa, da, ci, (H, W), k, s, d, p, bnb=(pre2, y2, coef2, True), gate=(gsrc, mask))
    out_b, part_b, _ = ops.conv_dgrad_f16x3(dy, w4d, wa, da, ci, (H, W), k, s, d, p, out=gated.clone(), accumulate=True, bnb=(pre2, y2, coef2, True))
    assert torch.equal(out_a, out_b) and torch.equal(part_a, part_b)
    # the residual layer's gate taken from bn_apply's bitmask of y instead of y itself (pfst_bnb_fuse_t.y_mask): the same bits, the same sums
    y2m, mask2 = ops.bn_apply(pre2, m2, i2, one, zero, True, residual=torch.randn(x.shape, generator=g(14)).to(DEV), want_mask=True)
    assert mask2 is not None and torch.equal(y2m, y2)
    out_c, part_c, _ = ops.conv_dgrad_f16x3(dy, w4d, wa, da, ci, (H, W), k, s, d, p, bnb=(pre2, y2, coef2, True, mask2), gate=(gsrc, mask))
    out_d, part_d, _ = ops.conv_dgrad_f16x3(dy, w4d, wa, da, ci, (H, W), k, s, d, p, bnb=(pre2, y2, coef2, True, mask2))
    out_e, part_e, _ = ops.conv_dgrad_f16x3(dy, w4d, wa, da, ci, (H, W), k, s, d, p, bnb=(pre2, y2, coef2, True))
    assert torch.equal(out_c, out_a) and torch.equal(part_c, part_a)
    assert torch.equal(out_d, out_e) and torch.equal(part_d, part_e)


@pytest.mark.parametrize('spread', [0.0, 2.5])
@pytest.mark.parametrize('case', [(2, 64, 64, 32, 32, 3, 1), (1, 32, 64, 20, 32, 3, 1), (2, 32, 32, 16, 48, 3, 2), (1, 64, 160, 12, 16, 3, 4),
                                  (2, 256, 64, 16, 16, 1, 1), (1, 48, 24, 10, 12, 1, 1), (3, 16, 40, 9, 16, 3, 1)])
def test_wgrad_f16x3_on_the_quad_kernel(ops, case, spread):
    """pfst_conv_wgrad_f16x3_q: the weight gradient of the direct stride-1 3x3 layers (stems, layer1) and of 1x1 layers with <= 64 output
    channels with the f16x3 split on the K-quad kernel -- interior / edge K-steps, dilations, ragged channel counts, accumulation; as close
    to fp64 as the whole-line f16x3 kernel is held (3e-6), on wide-range operands too."""
    n, ci, co, H, W, k, d = case
    p = d if k == 3 else 0
    x = torch.randn(n, ci, H, W, generator=g(1))
    dy = torch.randn(n, co, H, W, generator=g(4))
    if spread:
        x = x * 3e4 * torch.exp(spread * torch.randn(x.shape, generator=g(7)))
        dy = dy * 1e-6 * torch.exp(spread * torch.randn(dy.shape, generator=g(9)))
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, k, k), dy.double(), 1, p, d)
    xd, dyd = x.to(DEV), dy.to(DEV)
    dw = torch.zeros(co, ci, k, k, device=DEV)
    ops.conv_wgrad_f16q_(dw, xd, dyd, ops.absmax(xd), ops.absmax(dyd), k, d)
    assert_close(dw, dw_ref, 3e-6, 'f16x3 quad wgrad')
    e16 = rel_err(dw, dw_ref)
    ops.conv_wgrad_f16q_(dw, xd, dyd, ops.absmax(xd), ops.absmax(dyd), k, d)                 # accumulates (fp32 atomics)
    assert_close(dw, 2 * dw_ref, 3e-6, 'f16x3 quad wgrad accumulate')
    ref32 = torch.zeros(co, ci, k, k, device=DEV)
    ops.conv_wgrad_(ref32, xd, dyd, k, 1, d, p)                                               # the fp32-input MFMA kernel on the same data, for scale:
    assert e16 <= 3 * rel_err(ref32, dw_ref) + 2e-7                                           # both a few 1e-7 (measured 0.8e-7 fp32, 1-2e-7 f16x3)


def test_f16x3_refuses_shapes_it_does_not_cover(ops):
    from pfst_amd._lib import PfstHipError
    x = torch.randn(1, 48, 8, 8, device=DEV)
    w = torch.randn(128, 48, 3, 3, device=DEV)           # a 3x3 contraction over 1.5 channel blocks per tap
    w4f, _, wa = ops.pack_weight_f16x2(w, True, False)
    assert not ops.f16x3_eligible(48, 128) and not ops.f16x3_eligible(64, 32) and ops.f16x3_eligible(64, 96)
    assert ops.f16x3_eligible(64, 64) == (ops.F16X3_MIN_ROWS == 32)           # 33 ... 64 rows: the 64-row tile
    assert ops.f16x3_eligible(48, 128, 1) and not ops.f16x3_eligible(40, 128, 1)
    with pytest.raises((PfstHipError, AssertionError)):
        ops.conv_fprop_f16x3(x, w4f, wa, ops.absmax(x), 128, 3, pad=1)


def test_absmax_slot_groups(ops):
    """pfst_absmax: exact maxima per plane, channel slices of a concat buffer into one group, extension of an existing group, zeros"""
    t = torch.randn(3, 40, 9, 11, generator=g(3)).to(DEV)
    a = ops.absmax(t)
    assert a.numel() == ops.AMAX_SUB and float(a.max()) == float(t.abs().max())
    planes = ops.absmax(t, planes=3)
    assert [float(planes[i * ops.AMAX_SUB:(i + 1) * ops.AMAX_SUB].max()) for i in range(3)] == [float(t[i].abs().max()) for i in range(3)]
    sl = t[:, 8:24]                                                                  # dense planes, batch stride of the parent
    assert float(ops.absmax(sl).max()) == float(sl.abs().max())
    ext = ops.absmax(t[:1].contiguous())
    ops.absmax(t[1:].contiguous() * 3.0, out=ext)
    assert float(ext.max()) == max(float(t[:1].abs().max()), float((t[1:] * 3.0).abs().max()))
    assert float(ops.absmax(torch.zeros(2, 4, 4, 4, device=DEV)).max()) == 0.0


@pytest.mark.parametrize('m', [2, 4])
@pytest.mark.parametrize('case', [(1, 96, 128, 16, 16, 1), (2, 128, 96, 12, 20, 2), (1, 128, 160, 16, 24, 4), (2, 96, 96, 9, 13, 1)])
def test_winograd_on_the_f16x3_gemm(ops, case, m):
    """The Winograd pipeline with the transform-domain GEMMs and weight-gradient products on the f16x3 kernels: filter sets scaled per
    transform index, V / dM scaled by the maxima their transforms publish."""
    n, ci, co, H, W, d = case
    x = torch.randn(n, ci, H, W, generator=g(1))
    w = torch.randn(co, ci, 3, 3, generator=g(2)) * 0.1
    dy = torch.randn(n, co, H, W, generator=g(4)) * 1e-5
    ref = F.conv2d(x.double(), w.double(), None, 1, d, d)
    dx_ref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), 1, d, d)
    uf, ud, af, ad = ops.wino_pack_weight_f16(w.to(DEV), m=m)
    assert uf.dtype == torch.uint8 and af.numel() == (m + 2) ** 2 * ops.AMAX_SUB
    y, (v, v_amax) = ops.wino_conv(x.to(DEV), uf, co, d, m=m, u_amax=af, keep_v=True)
    assert_close(y, ref, WINO_TOL[m], 'winograd/f16x3 fprop')
    assert_close(ops.wino_conv(dy.to(DEV), ud, ci, d, m=m, u_amax=ad), dx_ref, WINO_TOL[m], 'winograd/f16x3 dgrad')
    if ops.wino_tiles(H, W, d, m) % 4 == 0:
        dw_ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, 3, 3), dy.double(), 1, d, d)
        dw = torch.zeros(co, ci, 3, 3, device=DEV)
        ops.wino_wgrad_(dw, x.to(DEV), dy.to(DEV), d, v=v, m=m, split=2, v_amax=v_amax)
        assert_close(dw, dw_ref, 2 * WINO_TOL[m], 'winograd/f16x3 wgrad from the kept V')
        dw2 = torch.zeros(co, ci, 3, 3, device=DEV)
        ops.wino_wgrad_(dw2, x.to(DEV), dy.to(DEV), d, m=m, split=2)
        assert_close(dw2, dw_ref, 2 * WINO_TOL[m], 'winograd/f16x3 wgrad')


def test_conv_channel_slice_views(ops):
    """conv reading / writing channel slices of bigger tensors (concat elimination)."""
    n, ci, co, H, W = 2, 32, 64, 10, 12
    big_in = torch.randn(n, ci + 16, H, W, generator=g(5)).to(DEV)
    w = (torch.randn(co, ci, 1, 1, generator=g(6)) * 0.1).to(DEV)
    big_out = torch.zeros(n, co + 8, H, W, device=DEV)
    wf, _ = ops.pack_weight(w, want_dgrad=False)
    ops.conv_fprop(big_in[:, 16:], wf, co, 1, out=big_out[:, 8:])
    ref = F.conv2d(big_in[:, 16:].cpu(), w.cpu())
    assert_close(big_out[:, 8:], ref, 2e-5)
    assert float(big_out[:, :8].abs().max()) == 0.0


# whole-plane kernel: dilation 1 (row_taps_d1), % 4 (16-byte taps), other, scalar; planes above 64 KB: strips + halo rows, same three tap forms
@pytest.mark.parametrize('dil,H,W', [(1, 16, 20), (12, 32, 32), (36, 24, 40), (2, 7, 9), (3, 24, 32), (1, 136, 128), (4, 130, 132), (2, 132, 128)])
def test_depthwise(ops, dil, H, W):
    n, c = 2, 24
    x = torch.randn(n, c, H, W, generator=g(1)).requires_grad_()
    w = torch.randn(c, 1, 3, 3, generator=g(2)).requires_grad_()
    y_ref = F.conv2d(x, w, None, 1, dil, dil, c)
    dy = torch.randn(y_ref.shape, generator=g(3))
    y_ref.backward(dy)
    xd, wd, dyd = x.detach().to(DEV), w.detach().to(DEV), dy.to(DEV)
    assert_close(ops.dwconv(xd, wd, dil), y_ref, 1e-5, 'dw fwd')
    y2, st, slots = ops.dwconv(xd, wd, dil, want_stats=True)            # fused BatchNorm statistics of the output
    mean, invstd = ops.bn_finalize_partials(st, slots, c, n * H * W)
    assert_close(y2, y_ref, 1e-5, 'dw fwd with stats')
    yr = y_ref.detach().double()
    assert_close(mean, yr.mean((0, 2, 3)), 2e-5, 'dw stats mean')
    assert_close(invstd, 1.0 / torch.sqrt(yr.var((0, 2, 3), unbiased=False) + 1e-5), 2e-5, 'dw stats invstd')
    assert_close(ops.dwconv(dyd, wd, dil, flip=True), x.grad, 1e-5, 'dw dgrad')
    acc = ops.dwconv(dyd, wd, dil, flip=True, out=xd.clone(), accumulate=True)
    assert_close(acc, x.grad + x.detach(), 1e-5, 'dw dgrad accumulate')
    dw = torch.zeros_like(wd)
    ops.dwconv_wgrad_(dw, xd, dyd, dil)
    assert_close(dw, w.grad, 1e-4, 'dw wgrad')
    # normalise-on-load (layers.conv_bn_act(defer=True)): the depthwise layer reads the PRE-BatchNorm tensor of the layer in front of it and
    # applies scale, shift and ReLU while staging -- forward and the fused backward's weight-gradient operand bit-identical to the same
    # kernels on the tensor bn_apply would have written
    pre = torch.randn(n, c, H, W, generator=g(11)).to(DEV) * 2
    gamma, beta = (torch.rand(c, generator=g(12)) + 0.5).to(DEV) * torch.where(torch.arange(c) % 5 == 0, -1.0, 1.0).to(DEV), torch.randn(c, generator=g(13)).to(DEV)
    mean, invstd, coef = ops.bn_stats(pre, gamma=gamma, beta=beta)
    ymat = ops.bn_apply(pre, mean, invstd, gamma, beta, True)
    assert torch.equal(ops.dwconv(pre, wd, dil, bnl=coef), ops.dwconv(ymat, wd, dil))
    ya, sta, sla = ops.dwconv(pre, wd, dil, want_stats=True, bnl=coef)
    yb, stb, slb = ops.dwconv(ymat, wd, dil, want_stats=True)
    assert torch.equal(ya, yb) and sla == slb
    dxa, dwa = torch.empty_like(xd), torch.zeros_like(wd)
    dxb, dwb = torch.empty_like(xd), torch.zeros_like(wd)
    ops.dwconv_bwd_(dwa, pre, dyd, wd, dil, dxa, bnl=coef)
    ops.dwconv_bwd_(dwb, ymat, dyd, wd, dil, dxb)
    assert torch.equal(dxa, dxb)
    assert_close(dwa, dwb, 1e-5, 'normalise-on-load: fused dw backward weight gradient')
    # BatchNorm backward's second pass applied on the fly (layers.FUSE_DW_BNBWD): dy is the gradient of THIS layer's BN + ReLU output, the
    # kernel stages dL/dpre = bn_backward(dy, pre) itself; against bn_backward writing dL/dpre first: dx bit-identical, parameter gradients equal
    pre2 = ops.dwconv(xd, wd, dil)
    g2, b2 = (torch.rand(c, generator=g(14)) + 0.5).to(DEV), torch.randn(c, generator=g(15)).to(DEV) * 0.3
    m2, i2, _ = ops.bn_stats(pre2, gamma=g2, beta=b2)
    dyo = torch.randn(n, c, H, W, generator=g(16)).to(DEV)
    dg_a, db_a, dg_b, db_b = (torch.zeros(c, device=DEV) for _ in range(4))
    dpre = ops.bn_backward(dyo, None, pre2, m2, i2, g2, dg_b, db_b, True, beta=b2)
    dxb, dwb = torch.empty_like(xd), torch.zeros_like(wd)
    ops.dwconv_bwd_(dwb, xd, dpre, wd, dil, dxb)
    rec = ops.bn_backward_sums(dyo, pre2, m2, i2, g2, b2, dg_a, db_a)
    dxa, dwa = torch.empty_like(xd), torch.zeros_like(wd)
    ops.dwconv_bwd_(dwa, xd, dyo, wd, dil, dxa, bnb=(pre2, rec))
    assert torch.equal(dxa, dxb), 'BN backward applied on the fly: dx'
    assert_close(dwa, dwb, 1e-5, 'BN backward applied on the fly: dw')
    assert_close(dg_a, dg_b, 1e-6, 'dgamma')
    assert_close(db_a, db_b, 1e-6, 'dbeta')
    # both gradients in one pass (pfst_dwconv3x3_bwd: the layers' backward since round 4): same dx as the data-gradient kernel bit for bit,
    # the weight gradient within fp32 summation order of the stand-alone kernel; accumulate variants of both outputs
    dx1, dw1 = torch.empty_like(xd), torch.zeros_like(wd)
    ops.dwconv_bwd_(dw1, xd, dyd, wd, dil, dx1)
    assert torch.equal(dx1, ops.dwconv(dyd, wd, dil, flip=True)), 'fused dw backward: dx'
    assert_close(dw1, w.grad, 1e-4, 'fused dw backward: dw')
    dx2 = xd.clone()
    ops.dwconv_bwd_(dw1, xd, dyd, wd, dil, dx2, accumulate=True)
    assert_close(dx2, x.grad + x.detach(), 1e-5, 'fused dw backward: dx accumulate')
    assert_close(dw1, 2 * w.grad, 1e-4, 'fused dw backward: dw accumulates')


@pytest.mark.parametrize('H,W,dils', [(32, 32, (12, 24, 36)), (128, 128, (12, 24, 36)), (24, 40, (4, 8)), (16, 16, (36,))])
def test_depthwise_branches_in_one_pass(ops, H, W, dils):
    """pfst_dwconv3x3_multi_fwd / _bwd (the ASPP head's atrous branches): every branch's output and BatchNorm partials bit-identical to
    its own pfst_dwconv3x3 launch; the backward's input gradient = the sum of the branches' data gradients, its weight gradients those of
    pfst_dwconv3x3_wgrad, both to fp32 summation order; accumulate into an existing input gradient."""
    n, c = 2, 20
    x = torch.randn(n, c, H, W, generator=g(1))
    ws = [torch.randn(c, 1, 3, 3, generator=g(2 + i)) for i in range(len(dils))]
    dys = [torch.randn(n, c, H, W, generator=g(7 + i)) for i in range(len(dils))]
    xd, wd, dyd = x.to(DEV), [w.to(DEV) for w in ws], [d.to(DEV) for d in dys]
    assert ops.dwconv_multi_ok(xd, list(dils)) and not ops.dwconv_multi_ok(xd, [3]) and not ops.dwconv_multi_ok(xd[:, :, :, 1:], list(dils))
    res = ops.dwconv_multi(xd, wd, list(dils), want_stats=True)
    for i, d in enumerate(dils):
        y1, st1, sl1 = ops.dwconv(xd, wd[i], d, want_stats=True)
        assert torch.equal(res[i][0], y1) and res[i][2] == sl1 and torch.equal(res[i][1][:2 * c * sl1], st1[:2 * c * sl1]), (i, d)
    dx_ref = sum(torch.nn.grad.conv2d_input(x.shape, ws[i].double(), dys[i].double(), 1, d, d, c) for i, d in enumerate(dils))
    dws, dx = [torch.zeros_like(w) for w in wd], torch.empty_like(xd)
    ops.dwconv_multi_bwd_(dws, xd, dyd, wd, list(dils), dx)
    assert_close(dx, dx_ref, 1e-5, 'multi-branch dw backward: dx')
    for i, d in enumerate(dils):
        dw_ref = torch.nn.grad.conv2d_weight(x.double(), ws[i].shape, dys[i].double(), 1, d, d, c)
        assert_close(dws[i], dw_ref, 1e-4, f'multi-branch dw backward: dw[{i}]')
    dx2 = xd.clone()
    ops.dwconv_multi_bwd_(dws, xd, dyd, wd, list(dils), dx2, accumulate=True)
    assert_close(dx2, dx_ref + x.double(), 1e-5, 'multi-branch dw backward: dx accumulate')
    # the image-pool branch's plane means from the same forward pass, their adjoint in the same backward pass
    res2, mean = ops.dwconv_multi(xd, wd, list(dils), want_stats=False, want_mean=True)
    assert torch.equal(res2[0][0], res[0][0])
    gap = ops.global_avgpool(xd)
    assert tuple(mean.shape) == (n, c, 1, 1) and float((mean - gap).abs().max()) <= 1e-7 * float(gap.abs().max())
    mg = torch.randn(n, c, generator=g(20))
    dx3 = torch.empty_like(xd)
    ops.dwconv_multi_bwd_([torch.zeros_like(w) for w in wd], xd, dyd, wd, list(dils), dx3, mean_grad=mg.to(DEV))
    assert_close(dx3, dx_ref + (mg.double() / (H * W)).view(n, c, 1, 1), 1e-5, 'multi-branch dw backward: + adjoint of the plane mean')
    # ... and with every branch's BatchNorm backward applied on the fly: the same as writing the three dL/dpre tensors first
    gam = [(torch.rand(c, generator=g(30 + i)) + 0.5).to(DEV) for i in range(len(dils))]
    bet = [(torch.randn(c, generator=g(40 + i)) * 0.3).to(DEV) for i in range(len(dils))]
    pres = [r_[0] for r_ in res]
    stats = [ops.bn_stats(pres[i], gamma=gam[i], beta=bet[i]) for i in range(len(dils))]
    dpres, recs = [], []
    for i in range(len(dils)):
        dpres.append(ops.bn_backward(dyd[i], None, pres[i], stats[i][0], stats[i][1], gam[i], None, None, True, beta=bet[i]))
        recs.append(ops.bn_backward_sums(dyd[i], pres[i], stats[i][0], stats[i][1], gam[i], bet[i], None, None))
    dws_a, dws_b = [torch.zeros_like(w) for w in wd], [torch.zeros_like(w) for w in wd]
    dxa, dxb = torch.empty_like(xd), torch.empty_like(xd)
    ops.dwconv_multi_bwd_(dws_b, xd, dpres, wd, list(dils), dxb)
    ops.dwconv_multi_bwd_(dws_a, xd, dyd, wd, list(dils), dxa, bnb=[(pres[i], recs[i]) for i in range(len(dils))])
    assert torch.equal(dxa, dxb), 'multi-branch dw backward with BN backward on the fly: dx'
    for i in range(len(dils)):
        assert_close(dws_a[i], dws_b[i], 1e-5, f'multi-branch dw backward with BN backward on the fly: dw[{i}]')


@pytest.mark.parametrize('shape,relu,res', [((4, 32, 16, 16), True, False), ((2, 48, 9, 13), True, True),
                                            ((3, 16, 8, 8), False, True), ((4, 64, 1, 1), True, False)])
def test_batchnorm_train(ops, shape, relu, res):
    n, c, h, w = shape
    x = (torch.randn(shape, generator=g(1)) * 2 + 0.5).requires_grad_()
    r = torch.randn(shape, generator=g(2)).requires_grad_() if res else None
    gamma = (torch.rand(c, generator=g(3)) + 0.5).requires_grad_()
    beta = torch.randn(c, generator=g(4)).requires_grad_()
    rm, rv = torch.zeros(c), torch.ones(c)
    y_ref = F.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 1e-5)
    if res:
        y_ref = y_ref + r
    if relu:
        y_ref = F.relu(y_ref)
    dy = torch.randn(shape, generator=g(5))
    y_ref.backward(dy)
    xd = x.detach().to(DEV)
    rmd, rvd = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    mean, invstd = ops.bn_stats(xd, rmd, rvd)
    assert_close(rmd, rm, 1e-5, 'running_mean')
    assert_close(rvd, rv, 1e-5, 'running_var')
    y = ops.bn_apply(xd, mean, invstd, gamma.detach().to(DEV), beta.detach().to(DEV), relu, None if r is None else r.detach().to(DEV))
    assert_close(y, y_ref, 1e-5, 'bn fwd')
    dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
    dres = torch.empty(shape, device=DEV) if res else None
    dx = ops.bn_backward(dy.to(DEV), y if res else None, xd, mean, invstd, gamma.detach().to(DEV), dg, db, relu, dres,
                         beta=beta.detach().to(DEV))
    assert_close(dx, x.grad, 1e-4, 'bn dx')
    assert_close(dg, gamma.grad, 1e-4, 'bn dgamma')
    assert_close(db, beta.grad, 1e-4, 'bn dbeta')
    if res:
        assert_close(dres, r.grad, 1e-6, 'bn dres')


@pytest.mark.parametrize('shape', [(2, 24, 16, 16), (3, 8, 32, 24), (1, 5, 16, 48)])
def test_batchnorm_relu_bitmask_equals_saved_output(ops, shape):
    """Residual BN+ReLU: the backward pass fed with the 1-bit ReLU gate of bn_apply gives bit-identical results to the one fed
    with the saved fp32 output (plane sizes are multiples of 256; a zero-crossing residual makes both gate values common),
    and the mask words hold exactly the bits (y > 0) in the documented layout."""
    n, c, h, w = shape
    x = (torch.randn(shape, generator=g(1)) * 2 + 0.5).to(DEV)
    r = torch.randn(shape, generator=g(2)).to(DEV)
    r[0, 0, 0, :8] = 0.0
    gamma, beta = (torch.rand(c, generator=g(3)) + 0.5).to(DEV), torch.randn(c, generator=g(4)).to(DEV)
    dy = torch.randn(shape, generator=g(5)).to(DEV)
    mean, invstd = ops.bn_stats(x)
    y, mask = ops.bn_apply(x, mean, invstd, gamma, beta, True, r, want_mask=True)
    assert mask is not None and mask.numel() == n * c * h * w // 64
    assert torch.equal(y, ops.bn_apply(x, mean, invstd, gamma, beta, True, r))
    # layout: 256-element chunk q of a plane -> 4 words; word k, bit l <-> element 256 q + 4 l + k
    gate = (y > 0).reshape(n * c, h * w // 256, 64, 4).permute(0, 1, 3, 2).cpu().numpy()          # [plane][q][k][l]
    words = mask.cpu().numpy().view(np.uint64).reshape(n * c, h * w // 256, 4)
    bits = (words[..., None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)
    assert np.array_equal(bits.astype(bool), gate)
    out = []
    for use_mask in (False, True):
        dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        dres = torch.full(shape, 0.25, device=DEV)
        dx = ops.bn_backward(dy, None if use_mask else y, x, mean, invstd, gamma, dg, db, True, dres, dres_accumulate=True, beta=beta,
                             mask=mask if use_mask else None)
        out.append((dx, dres, dg, db))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])        # dx, dres: element-wise, no reduction order
    assert_close(out[1][2], out[0][2], 1e-6, 'dgamma')
    assert_close(out[1][3], out[0][3], 1e-6, 'dbeta')
    # planes that are not a multiple of 256: no mask, the caller keeps y
    x2 = torch.randn(2, 4, 9, 13, generator=g(6)).to(DEV)
    m2, i2 = ops.bn_stats(x2)
    y2, none = ops.bn_apply(x2, m2, i2, gamma[:4], beta[:4], True, x2, want_mask=True)
    assert none is None


@pytest.mark.parametrize('H,W', [(17, 22), (16, 24), (2, 2), (64, 64), (6, 4), (128, 96)])      # odd sizes: generic kernels; even, W % 4 == 0: pair forward, 2x2-block backward
def test_maxpool(ops, H, W):
    x = torch.randn(2, 8, H, W, generator=g(1)).requires_grad_()
    with torch.no_grad():
        x[0, 0, :4, :4] = 1.5                       # ties: the first maximum in scan order must win, as in torch
    y_ref = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(y_ref.shape, generator=g(2))
    y_ref.backward(dy)
    y, idx = ops.maxpool(x.detach().to(DEV))
    assert torch.equal(y.cpu(), y_ref.detach())
    xn = x.detach().clone()
    xn[1, 3, H // 2, W // 2] = float('nan')          # a NaN tap wins its windows, as in torch
    yn, idn = ops.maxpool(xn.to(DEV))
    assert torch.equal(torch.isnan(yn).cpu(), torch.isnan(F.max_pool2d(xn, 3, 2, 1)))
    # the pair kernel (even sizes, W % 4 == 0) against the one-output kernel, reached through a 4-byte aligned view: same values, same taps
    if H % 2 == 0 and W % 4 == 0:
        pad = torch.zeros(2 * 8 * H * W + 1, device=DEV)
        pad[1:] = x.detach().to(DEV).reshape(-1)
        y1, idx1 = ops.maxpool(pad[1:].reshape(2, 8, H, W))
        assert torch.equal(y1, y) and torch.equal(idx1, idx)
    dx = ops.maxpool_bwd(dy.to(DEV), idx, (H, W))
    assert_close(dx, x.grad, 1e-6)
    # normalise-on-load (stem.6 -> max-pool): pooling the pre-BatchNorm tensor with (sc, sh, ReLU) applied per tap = pooling what bn_apply
    # would have written, values and winning taps alike (negative scales included: the pool is not taken before the affine map)
    pre = (torch.randn(2, 8, H, W, generator=g(3)) * 2).to(DEV)
    gamma = (torch.tensor([1.0, -0.7, 0.3, 2.0, -1.5, 0.9, 1.1, -0.2])).to(DEV)
    beta = torch.randn(8, generator=g(4)).to(DEV)
    mean, invstd, coef = ops.bn_stats(pre, gamma=gamma, beta=beta)
    ymat = ops.bn_apply(pre, mean, invstd, gamma, beta, True)
    y1, i1 = ops.maxpool(pre, bnl=coef)
    y0, i0 = ops.maxpool(ymat)
    assert torch.equal(y1, y0) and torch.equal(i1, i0)
    # max |y| published by the pooling launch itself (the f16x3 scale of layer1's GEMMs): the group's maximum is the tensor's, exactly
    for kw in ({}, {'bnl': coef}):
        slots = ops.amax_slots(pre.device)
        yk, _ = ops.maxpool(pre, amax=slots, **kw)
        assert slots.max().item() == yk.abs().max().item()


@pytest.mark.parametrize('hi,wi,ho,wo', [(8, 8, 16, 16), (16, 12, 64, 48), (5, 7, 13, 9), (1, 1, 6, 6), (16, 16, 128, 128),
                                         (6, 10, 12, 20), (2, 2, 4, 4), (33, 17, 66, 34), (32, 48, 64, 96), (128, 128, 256, 256)])   # from (6, 10): exact-2x fast paths
def test_bilinear_resize(ops, hi, wi, ho, wo):
    x = torch.randn(2, 5, hi, wi, generator=g(1)).requires_grad_()
    y_ref = F.interpolate(x, size=(ho, wo), mode='bilinear', align_corners=False)
    dy = torch.randn(y_ref.shape, generator=g(2))
    y_ref.backward(dy)
    y = ops.resize_bilinear(x.detach().to(DEV), (ho, wo))
    assert_close(y, y_ref, 1e-6, 'resize fwd')
    dx = ops.resize_bilinear_bwd(dy.to(DEV), (hi, wi))
    assert_close(dx, x.grad, 1e-5, 'resize bwd')
    dx2 = ops.resize_bilinear_bwd(dy.to(DEV), (hi, wi), out=dx.clone(), accumulate=True)
    assert_close(dx2, 2 * x.grad, 1e-5, 'resize bwd accumulate')
    if ho == 2 * hi and wo == 2 * wi and hi % 2 == 0 and wi % 2 == 0:
        # the 2 x 2-block adjoint against the one-pixel kernel (reached through an 8-byte aligned view): the same taps and weights, the
        # compiler contracts the two expressions into differently grouped fmas
        pad = torch.zeros(2, 5 * ho * wo + 2, device=DEV)
        pad[:, 2:] = dy.to(DEV).reshape(2, -1)
        dx1 = ops.resize_bilinear_bwd(pad[:, 2:].reshape(2, 5, ho, wo), (hi, wi))       # 8-byte aligned images: the one-pixel kernel
        assert_close(dx, dx1.double(), 1e-6, 'resize bwd, 2x2 blocks vs one pixel per thread')


def test_pool_broadcast_dropout(ops):
    x = torch.randn(3, 7, 9, 11, generator=g(1))
    xd = x.to(DEV)
    assert_close(ops.global_avgpool(xd), x.mean((2, 3), keepdim=True), 1e-6)
    assert_close(ops.reduce_hw(xd), x.sum((2, 3), keepdim=True), 1e-6)
    v = torch.randn(3, 7, 1, 1, generator=g(2))
    out = torch.zeros(3, 7, 9, 11, device=DEV)
    ops.broadcast_hw(v.to(DEV), out, 0.5)
    assert_close(out, (0.5 * v).expand(3, 7, 9, 11), 1e-6)
    slots = ops.amax_slots(out.device)
    ops.broadcast_hw(v.to(DEV), out, 0.5, amax=slots)
    assert slots.max().item() == out.abs().max().item()
    m = (torch.rand(3, 7, generator=g(3)) > 0.3).float() / 0.7
    assert_close(ops.channel_scale(xd, m.to(DEV)), x * m.view(3, 7, 1, 1), 1e-6)
    a, b = torch.randn(1000, generator=g(4)), torch.randn(1000, generator=g(5))
    ad = a.to(DEV)
    ops.axpy_(ad, b.to(DEV), 0.25)
    assert_close(ad, a + 0.25 * b, 1e-6)


@pytest.mark.parametrize('C,h,H,use_w,use_cw', [(6, 16, 64, True, False), (6, 8, 64, False, False),
                                                (33, 12, 48, True, True), (2, 16, 64, True, False), (6, 64, 256, True, True), (8, 32, 128, False, False),
                                                (6, 7, 28, False, True), (5, 31, 124, True, True), (6, 2, 8, True, False)])
def test_ce_upsample_fwd_bwd(ops, C, h, H, use_w, use_cw):
    n = 2
    logits = (torch.randn(n, C, h, h, generator=g(1)) * 3).requires_grad_()
    label = torch.randint(0, C, (n, H, H), generator=g(2))
    label[:, :5, :9] = 255
    pw = torch.rand(n, H, H, generator=g(3)) if use_w else None
    cw = (torch.rand(C, generator=g(4)) + 0.5) if use_cw else None
    up = F.interpolate(logits, size=(H, H), mode='bilinear', align_corners=False)
    per = F.cross_entropy(up, label, weight=cw, reduction='none', ignore_index=255)
    if pw is not None:
        per = per * pw
    loss_ref = 0.4 * per.mean()
    loss_ref.backward()
    keep = label != 255
    acc_ref = 100.0 * ((up.argmax(1) == label) & keep).sum().item() / keep.sum().item()
    ld = logits.detach().to(DEV)
    l8 = ops.to_u8(label.to(DEV))
    pwd = None if pw is None else pw.to(DEV)
    cwd = None if cw is None else cw.to(DEV)
    lse, acc = ops.ce_upsample_fwd(ld, l8, pwd, cwd)
    out = ops.ce_finalize(acc, n * H * H, 0.4).cpu()
    assert abs(float(out[0]) - float(loss_ref)) < 1e-5 * max(1.0, abs(float(loss_ref)))
    assert abs(float(out[1]) - acc_ref) < 1e-3
    dl = ops.ce_upsample_bwd(ld, l8, lse, 0.4 / (n * H * H), pwd, cwd)
    assert_close(dl, logits.grad, 1e-4, 'ce bwd')


@pytest.mark.parametrize('S', [4, 8])
@pytest.mark.parametrize('C,h,w', [(6, 33, 17), (8, 16, 46), (1, 15, 15), (6, 30, 31), (3, 1, 9)])
def test_ce_inter_cell_block_kernels_match_the_per_pixel_and_per_cell_kernels(ops, C, h, w, S):
    """H == S h, W == S w (S = 4: the decode head, 8: the auxiliary head): the inter-cell-block kernels (ce_fwd_blocks_kernel,
    ce_bwd_blocks_kernel) against autograd AND against the kernels they replace (taken by giving the label map an odd address, which the block
    kernels' 2-byte label loads refuse): forward bit-identical, backward the same terms in another summation order"""
    n, H, W = 3, S * h, S * w
    logits = (torch.randn(n, C, h, w, generator=g(1)) * 3).requires_grad_()
    label = torch.randint(0, C + 1, (n, H, W), generator=g(2))
    label[label == C] = 255
    label[:, :3, :] = 255
    pw = torch.rand(n, H, W, generator=g(3))
    cw = torch.rand(C, generator=g(4)) + 0.5
    up = F.interpolate(logits, size=(H, W), mode='bilinear', align_corners=False)
    per = F.cross_entropy(up, label, weight=cw, reduction='none', ignore_index=255) * pw
    (per.sum() * 0.01).backward()
    ld, pwd, cwd = logits.detach().to(DEV), pw.to(DEV), cw.to(DEV)
    l8 = ops.to_u8(label.to(DEV))
    odd = torch.empty(l8.numel() + 1, dtype=torch.uint8, device=DEV)[1:].view(n, H, W)
    odd.copy_(l8)
    assert l8.data_ptr() % 2 == 0 and odd.data_ptr() % 2 == 1
    lse, acc = ops.ce_upsample_fwd(ld, l8, pwd, cwd)
    # forward: the block kernel evaluates every pixel exactly as the one-pixel-per-thread kernel does
    lse_px, acc_px = ops.ce_upsample_fwd(ld, odd, pwd, cwd)
    assert torch.equal(lse, lse_px), 'x4 forward: lse must be bit-identical to the per-pixel kernel'
    assert torch.equal(acc[1:], acc_px[1:]) and abs(float(acc[0] - acc_px[0])) <= 1e-12 * abs(float(acc_px[0]))
    assert abs(float(acc[0]) * 0.01 - float(per.sum() * 0.01)) < 1e-5 * max(1.0, abs(float(per.sum() * 0.01)))
    dl = ops.ce_upsample_bwd(ld, l8, lse, 0.01, pwd, cwd)
    dl_cells = ops.ce_upsample_bwd(ld, odd, lse, 0.01, pwd, cwd)
    assert_close(dl, logits.grad, 1e-4, 'ce bwd x4 vs autograd')
    assert_close(dl, dl_cells, 2e-6, 'ce bwd x4 vs the per-cell gather')
    # accumulate form, and no pixel / class weights
    base = torch.randn(n, C, h, w, generator=g(5)).to(DEV)
    acc = ops.ce_upsample_bwd(ld, l8, lse, 0.01, pwd, cwd, out=base.clone(), accumulate=True)
    assert_close(acc, base + dl, 1e-6, 'ce bwd x4 accumulate')
    lse2, _ = ops.ce_upsample_fwd(ld, l8)
    assert_close(ops.ce_upsample_bwd(ld, l8, lse2, 0.01), ops.ce_upsample_bwd(ld, odd, lse2, 0.01), 2e-6, 'ce bwd x4, unweighted')


def test_pseudo_label_bit_exact(ops):
    n, C, h, H = 2, 6, 32, 128
    logits = torch.randn(n, C, h, h, generator=g(1)) * 4
    up = F.interpolate(logits, size=(H, H), mode='bilinear', align_corners=False)
    prob, lab = torch.max(torch.softmax(up, 1), 1)
    l64, l8, cnt = ops.pseudo_label(logits.to(DEV), (H, H), 0.9)
    assert torch.equal(l64.cpu(), lab), 'pseudo-label index map must be bit exact'
    assert torch.equal(l8.cpu().long(), lab)
    assert abs(int(cnt.item()) - int((prob >= 0.9).sum())) <= 2   # 1-ulp threshold ties only


def _near_tie_logits(C, h, seed, mags):
    """logits whose two best classes are 0, 1 or 2 ulp apart, in both index orders, at the given magnitudes; a quarter of the
    pixels have ALL classes equal.  (flat early-training teacher logits produce exactly these situations)"""
    gen = g(seed)
    n = 2
    base = torch.randn(n, C, h, h, generator=gen) * 0.5
    t = torch.tensor(mags)[torch.randint(0, len(mags), (n, 1, h, h), generator=gen)]
    a = torch.randint(0, C, (n, 1, h, h), generator=gen)
    b = (a + torch.randint(1, C, (n, 1, h, h), generator=gen)) % C            # b < a and b > a both occur
    k = torch.randint(0, 4, (n, 1, h, h), generator=gen)
    z = (t - 3.0 - base.abs()).clone()
    z.scatter_(1, a, t)
    zb = t.clone()
    down = t
    for kk in (1, 2):
        down = torch.nextafter(down, torch.full_like(down, -float('inf')))
        zb = torch.where(k == kk, down, zb)
    z.scatter_(1, b, zb)
    return torch.where(k == 3, t.expand_as(z), z).contiguous(), k


@pytest.mark.parametrize('C', [2, 6, 33])
@pytest.mark.parametrize('up', [1, 4])
def test_pseudo_label_tie_rule_is_softmax_then_max(ops, C, up):
    """pfgst.py:259-261: `torch.max(torch.softmax(logits, 1), 1)` -- the FIRST class whose rounded PROBABILITY is maximal, which
    is not the arg-max of the logits when distinct logits give equal probabilities.  Adversarial near-tie logits (top two 0 / 1 / 2
    ulp apart in both index orders, all-equal pixels), without and with the fused bilinear up-sampling.
    Bit-exact against torch's own softmax->max evaluated with the device's exp (the arithmetic the kernel pins: sequential fp32
    sum, IEEE division; probabilities agree to 0 ulp).  Against torch-CPU the map may differ ONLY where torch's CPU and GPU builds
    differ from each other (the CPU softmax uses Sleef's 2-ulp vector exp; measured: <= 8 of 8192 adversarial pixels)."""
    h = 32
    for mags in ([0.01, 0.1, 0.5], [1.0, 3.0, 7.5], [-0.02, -1.0, -6.0], [0.003, 20.0, -40.0]):
        z, k = _near_tie_logits(C, h, 11 + C, mags)
        H = h * up
        zu = z if up == 1 else F.interpolate(z, size=(H, H), mode='bilinear', align_corners=False)
        if up > 1:
            mine = ops.resize_bilinear(z.to(DEV), (H, H)).cpu()
            assert torch.equal(mine, zu), 'bilinear up-sampling must reproduce F.interpolate bit for bit (power-of-two scale)'
        p_cpu, l_cpu = torch.max(torch.softmax(zu, 1), 1)
        p_dev, l_dev = torch.max(torch.softmax(zu.to(DEV), 1), 1)
        l64, l8, cnt, prob = ops.pseudo_label(z.to(DEV), (H, H), 0.5, want_prob=True)
        assert torch.equal(l64.cpu(), l_dev.cpu()), f'C={C} mags={mags}: label map differs from torch.max(torch.softmax) on {int((l64.cpu() != l_dev.cpu()).sum())} pixels'
        assert torch.equal(prob.cpu(), p_dev.cpu()), 'max probability must equal torch softmax bit for bit'
        assert int(cnt.item()) == int((p_dev >= 0.5).sum())
        diff_cpu = l64.cpu() != l_cpu
        assert bool((diff_cpu <= (l_dev.cpu() != l_cpu)).all()) and int(diff_cpu.sum()) <= 16
        if up == 1:
            assert bool((l64.cpu()[k[:, 0] == 3] == 0).all()), 'all-equal logits: the first class wins'
            # the rule matters: the arg-max of the logits is a different map on these inputs
            if mags[0] in (0.01, -0.02):
                assert int((z.argmax(1) != l_cpu).sum()) > 100


@pytest.mark.parametrize('hi,ho', [(32, 128), (16, 128), (33, 100), (24, 96), (64, 256), (96, 192), (128, 256), (34, 68), (66, 132)])
def test_resize_bilinear_bit_exact(ops, hi, ho):
    """the up-sampling arithmetic is pinned to torch's (source index = one fma; blend = fma(lx0, v00, lx1*v01), fma(ly0, t0, ly1*t1)):
    what torch's GPU kernel and its vectorised CPU kernel evaluate.  (torch-CPU switches to a differently rounded scalar loop for
    output widths <= 64 -- measured, 46 % of the elements differ by an ulp there between torch's OWN two paths; every up-sampling
    on the PFST path writes 256..1024-wide planes.)"""
    x = torch.randn(2, 5, hi, hi, generator=g(hi)) * 3
    ref = F.interpolate(x, size=(ho, ho), mode='bilinear', align_corners=False)
    assert torch.equal(ops.resize_bilinear(x.to(DEV), (ho, ho)).cpu(), ref)
    # into a channel slice of a concat buffer (the decoder: the x2 kernel's 16-byte stores on a strided batch)
    cat = torch.zeros(2, 9, ho, ho, device=DEV)
    ops.resize_bilinear(x.to(DEV), (ho, ho), out=cat[:, 0:5])
    assert torch.equal(cat[:, 0:5].cpu(), ref) and float(cat[:, 5:].abs().max()) == 0.0


def test_class_mix_exact(ops):
    n, S = 3, 32
    gt = torch.randint(0, 6, (n, 1, S, S), generator=g(1))
    gt[:, :, :3, :3] = 255
    gt8 = ops.to_u8(gt.to(DEV))
    pres = ops.label_presence(gt8).cpu()
    assert sorted(torch.nonzero(pres).flatten().tolist()) == sorted(torch.unique(gt).tolist())
    classes = torch.tensor([[0, 3, 255, -1], [1, 2, -1, -1], [5, 4, 0, 1]], dtype=torch.int32)
    mask = ops.class_mask(gt8, classes.to(DEV))
    ref_mask = torch.zeros_like(gt)
    for i in range(n):
        for c in classes[i].tolist():
            if c >= 0:
                ref_mask[i] |= (gt[i] == c).long()
    assert torch.equal(mask.cpu().long(), ref_mask)
    img, trg = torch.randn(n, 3, S, S, generator=g(2)), torch.randn(n, 3, S, S, generator=g(3))
    pl = torch.randint(0, 6, (n, S, S), generator=g(4))
    cnt = torch.tensor([777], dtype=torch.int64)
    q = 777 / (n * S * S)
    mi, ml, ml64, mw = ops.class_mix(img.to(DEV), trg.to(DEV), gt8, ops.to_u8(pl.to(DEV)), mask, cnt.to(DEV), want_i64=True)
    m = ref_mask
    assert torch.equal(mi.cpu(), m * img + (1 - m) * trg)
    assert torch.equal(ml64.cpu(), m * gt + (1 - m) * pl.unsqueeze(1))
    assert torch.equal(ml.cpu().long(), ml64.cpu())
    assert torch.equal(mw.cpu(), m[:, 0] * torch.ones(n, S, S) + (1 - m[:, 0]) * (q * torch.ones(n, S, S)))


def test_pfgst_loss_pieces_against_golden(ops, golden_dir):
    """Whole PFGSTLoss (values + both gradients) against the vectors the reference produced."""
    import os
    s = np.load(os.path.join(golden_dir, 'small_ops.npz'))
    lt = torch.from_numpy(s['pl_logits_trg']).to(DEV)
    xe = torch.from_numpy(s['pl_x_ema']).to(DEV)
    xs = torch.from_numpy(s['pl_x_src']).to(DEV)
    gt8 = ops.to_u8(torch.from_numpy(s['pl_gt_src']).to(DEV))
    mm8 = ops.to_u8(torch.from_numpy(s['pl_mix_masks']).to(DEV))
    W = 0.1
    ema_sim, _ = ops.sim_map(xe, 2)
    src_sim, src_norm = ops.sim_map(xs, 2)
    l4, gsim = ops.src_sim_losses(src_sim, gt8, 2, W, W, W, W)
    dxs = ops.sim_map_bwd(xs, src_sim, src_norm, gsim, 2)
    prob = ops.softmax_down(lt, 2)
    valid, all9, cnt = ops.trg_valid_mask(gt8, mm8, prob.shape[-2:], 2)
    l2, gP = ops.sim_topk_loss(ema_sim, prob, valid, cnt, 2, 3, W, W)
    dl = torch.zeros_like(lt)
    ops.cross_prob_bwd_(dl, prob, gP, 2, 2)
    got = torch.cat([l4, l2]).cpu().double().numpy()
    assert np.allclose(got, s['pl_losses'], rtol=1e-4, atol=1e-7), (got, s['pl_losses'])
    assert np.array_equal(all9.cpu().numpy().astype(bool), s['pl_vis_mask'])
    dens = 1 - ema_sim.mean(1, keepdim=True)
    assert_close(dens, torch.from_numpy(s['pl_vis_density']), 1e-4, 'density')
    assert_close(dxs, torch.from_numpy(s['pl_grad_xsrc']), 1e-3, 'd x_src')
    assert_close(dl, torch.from_numpy(s['pl_grad_logits']), 1e-3, 'd logits_trg')


@pytest.mark.parametrize('n,C,H,W,d', [(2, 24, 16, 64, 2), (1, 36, 8, 128, 1), (2, 8, 4, 256, 2), (1, 512, 8, 128, 2), (2, 20, 6, 64, 1),
                                       (2, 12, 10, 20, 2)])       # the last shape takes the generic kernel
def test_sim_map_fast_path_against_torch(ops, n, C, H, W, d):
    """cosine similarity to the 9 dilated neighbours (pfgst_loss.py:193-208) and its adjoint: the strip kernels (16-byte loads, taps from
    the neighbouring lanes, channel quarters combined through LDS) against F.unfold + F.cosine_similarity + autograd."""
    x = torch.randn(n, C, H, W, generator=g(C + W)).requires_grad_()
    u = F.unfold(x, 3, dilation=d, padding=d).view(n, C, 9, H, W)
    ref = F.cosine_similarity(u, x.unsqueeze(2), dim=1)
    gs = torch.randn(n, 9, H, W, generator=g(7))
    (ref * gs).sum().backward()
    xd = x.detach().to(DEV)
    sim, norm = ops.sim_map(xd, d)
    assert float((sim.cpu() - ref.detach()).abs().max()) < 2e-6
    assert_close(norm, x.detach().norm(dim=1), 1e-6, 'feature norm')
    dx = ops.sim_map_bwd(xd, sim, norm, gs.to(DEV), d)
    assert_close(dx, x.grad, 1e-5, 'sim_map adjoint')
    base = torch.randn(n, C, H, W, generator=g(9)).to(DEV)
    acc = ops.sim_map_bwd(xd, sim, norm, gs.to(DEV), d, out=base.clone(), accumulate=True)
    assert_close(acc - base, x.grad, 1e-5, 'sim_map adjoint, accumulate')


def test_ema_adamw_flat(ops):
    n = 10007
    p = torch.randn(n, generator=g(1)).requires_grad_()
    t = torch.randn(n, generator=g(2))
    grad = torch.randn(n, generator=g(3))
    opt = torch.optim.AdamW([p], lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
    pd, td = p.detach().clone().to(DEV), t.clone().to(DEV)
    # EMA buffers must be 16-byte aligned: fresh allocations are
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in (1, 2, 3):
        p.grad = grad * step
        opt.step()
        ops.adamw_step_(pd, (grad * step).to(DEV), m, v, 6e-5, (0.9, 0.999), 1e-8, 0.01, step)
    assert_close(pd, p, 1e-6, 'adamw')
    ops.ema_update_(td, pd, 0.999)
    assert_close(td, 0.999 * t + (1 - 0.999) * p.detach(), 1e-6, 'ema')


PFGST_VARIANTS = {          # the same table as tests/test_oracle_golden.py / tests/golden/make_golden.py
    'gaussian': dict(sim_type='gaussian', sigma=8.0),
    'margin': dict(src_loss_type='margin', margin=(0.5, 0.1)),
    'margin2': dict(src_loss_type='margin2', margin=(0.6, 0.0)),
    'unfold_grad': dict(detach_unfold=False),
    'all_pairs': dict(top_k=None),
    'full_res': dict(downscale=None),
    'gaussian_all_unfold': dict(sim_type='gaussian', sigma=6.0, top_k=None, detach_unfold=False, src_loss_type='margin2',
                                margin=(0.7, 0.2)),
}


@pytest.mark.parametrize('name', list(PFGST_VARIANTS))
def test_pfgst_loss_option_variants_against_golden(ops, golden_dir, name):
    """The PFGSTLoss MODULE (host mirror + HIP kernels) under the option variants of the reference, values and both
    gradients against vectors produced by the executed reference (tests/golden/pfgst_options.npz)."""
    import os
    import pfst_amd  # noqa: F401
    from pfst_amd.engine import Tape, Var
    from pfst_amd.uda import PFGSTLoss
    z = np.load(os.path.join(golden_dir, 'pfgst_options.npz'))
    cfg = dict(kernel_size=3, dilation=2, top_k=3, weights={k: 0.1 for k in ('src_pos', 'src_neg', 'sim_pos', 'sim_neg', 'src_pos_std', 'src_neg_std')},
               sim_type='cosine', feat_level=None, detach_unfold=True, downscale=0.5)
    cfg.update(PFGST_VARIANTS[name])
    loss = PFGSTLoss(**cfg)
    lt = Var(torch.from_numpy(z['logits_trg']).to(DEV), True)
    xs = Var(torch.from_numpy(z['x_src']).to(DEV), True)
    xe = Var(torch.from_numpy(z['x_ema']).to(DEV), False)
    tape = Tape()
    out = loss(dict(logits_trg=lt, x_ema=xe, x_src=xs, gt_src=ops.to_u8(torch.from_numpy(z['gt_src']).to(DEV)),
                    mix_masks=ops.to_u8(torch.from_numpy(z['mix_masks']).to(DEV)), want_vis=True), tape)
    names = [k for k in out if not k.startswith('vis|')]
    assert names == list(z[name + '|names'])
    got = np.array([float(out[k].sum()) for k in names])
    assert np.allclose(got, z[name + '|losses'], rtol=1e-4, atol=1e-7), (got, z[name + '|losses'])
    tape.backward()
    assert_close(lt.grad, torch.from_numpy(z[name + '|grad_logits']), 1e-3, 'd logits_trg')
    assert_close(xs.grad, torch.from_numpy(z[name + '|grad_xsrc']), 1e-3, 'd x_src')
    dens = out['vis|density_sim_feat'][1]          # the reference's tuple: (img_trg, 1 - mean_k sim_ema, unmixed-neighbourhood mask)
    assert_close(dens, torch.from_numpy(z[name + '|density']), 1e-4, 'density')


PFGST_VARIANTS2 = {         # tests/golden/make_golden.py:PFGST_OPTION_VARIANTS2 (src_perc, proj_net -- SURVEY.md §8 f4)
    'src_perc': dict(src_perc=0.6),
    'src_perc_margin': dict(src_perc=0.35, src_loss_type='margin', margin=(0.5, 0.1)),
    'proj_net': dict(proj_net_cfg=dict(in_channels=32, out_channels=16)),
    'proj_net_src_perc_all': dict(proj_net_cfg=dict(in_channels=32, out_channels=24), src_perc=0.8, top_k=None),
}


@pytest.mark.parametrize('name', list(PFGST_VARIANTS2))
def test_pfgst_loss_src_perc_and_proj_net_against_golden(ops, golden_dir, name):
    """src_perc (radix-select threshold instead of the reference's sort, pfgst_loss.py:98-102) and proj_net (trainable 1x1
    projection of both feature maps, :34-36,73-75; its weights also receive the gradient of the teacher-side similarity):
    loss values, both input gradients and the projection's gradients against vectors from the executed reference."""
    import os
    import pfst_amd  # noqa: F401
    from pfst_amd.engine import Tape, Var
    from pfst_amd.uda import PFGSTLoss
    z = np.load(os.path.join(golden_dir, 'pfgst_options2.npz'))
    cfg = dict(kernel_size=3, dilation=2, top_k=3, weights={k: 0.1 for k in ('src_pos', 'src_neg', 'sim_pos', 'sim_neg', 'src_pos_std', 'src_neg_std')},
               sim_type='cosine', feat_level=None, detach_unfold=True, downscale=0.5)
    cfg.update(PFGST_VARIANTS2[name])
    loss = PFGSTLoss(**cfg).to(DEV)
    if loss.proj_net is not None:
        with torch.no_grad():
            loss.proj_net.weight.copy_(torch.from_numpy(z[name + '|proj_weight']))
            loss.proj_net.bias.copy_(torch.from_numpy(z[name + '|proj_bias']))
    lt = Var(torch.from_numpy(z['logits_trg']).to(DEV), True)
    xs = Var(torch.from_numpy(z['x_src']).to(DEV), True)
    xe = Var(torch.from_numpy(z['x_ema']).to(DEV), False)
    tape = Tape()
    out = loss(dict(logits_trg=lt, x_ema=xe, x_src=xs, gt_src=ops.to_u8(torch.from_numpy(z['gt_src']).to(DEV)),
                    mix_masks=ops.to_u8(torch.from_numpy(z['mix_masks']).to(DEV)), want_vis=True), tape)
    names = [k for k in out if not k.startswith('vis|')]
    assert names == list(z[name + '|names'])
    got = np.array([float(out[k].sum()) for k in names])
    assert np.allclose(got, z[name + '|losses'], rtol=1e-4, atol=1e-7), (got, z[name + '|losses'])
    tape.backward()
    assert_close(lt.grad, torch.from_numpy(z[name + '|grad_logits']), 1e-3, 'd logits_trg')
    assert_close(xs.grad, torch.from_numpy(z[name + '|grad_xsrc']), 1e-3, 'd x_src')
    assert_close(out['vis|density_sim_feat'][1], torch.from_numpy(z[name + '|density']), 1e-4, 'density')
    if loss.proj_net is not None:
        assert_close(loss.proj_net.weight.grad, torch.from_numpy(z[name + '|grad_proj_weight']), 1e-3, 'd proj_net.weight')
        assert_close(loss.proj_net.bias.grad, torch.from_numpy(z[name + '|grad_proj_bias']), 1e-3, 'd proj_net.bias')


def test_src_perc_selection_counts(ops):
    """the radix select keeps exactly int(n * perc) pairs per set (ties weighted fractionally): the weighted counts equal the
    reference's prefix lengths, for several fractions incl. the degenerate ones"""
    gen = g(3)
    n, H = 2, 16
    sim = (torch.rand(n, 9, H, H, generator=gen) * 2 - 1).round(decimals=2)          # many exact ties
    gt = torch.randint(0, 4, (n, 1, 4, 4), generator=gen).repeat_interleave(H // 4 * 8, 2).repeat_interleave(H // 4 * 8, 3)
    gt[:, :, :8, :8] = 255
    gt8 = ops.to_u8(gt.to(DEV))
    _, g_all = ops.src_sim_losses(sim.to(DEV), gt8, 2, 0.1, 0.1, 0.1, 0.1)
    for perc in (1.0, 0.5, 0.123, 0.0):
        from pfst_amd._lib import lib
        sel = torch.empty(lib().pfst_src_sim_select_bytes() // 8 + 1, dtype=torch.int64, device=DEV)
        hg = gt.shape[-1]
        ops.call('pfst_src_sim_select', sim.to(DEV).data_ptr(), gt8.data_ptr(), n, H, H, hg, hg, 2, float(perc), sel.data_ptr(), 0)
        stats = torch.zeros(6, dtype=torch.float64, device=DEV)
        ops.call('pfst_src_sim_stats', sim.to(DEV).data_ptr(), gt8.data_ptr(), n, H, H, hg, hg, 2, 0, 0.5, 0.5, stats.data_ptr(), sel.data_ptr(), 0)
        full = torch.zeros(6, dtype=torch.float64, device=DEV)
        ops.call('pfst_src_sim_stats', sim.to(DEV).data_ptr(), gt8.data_ptr(), n, H, H, hg, hg, 2, 0, 0.5, 0.5, full.data_ptr(), 0, 0)
        torch.cuda.synchronize()
        for o in (0, 3):
            assert abs(float(stats[o]) - int(float(full[o]) * perc)) < 1e-3, (perc, o, float(stats[o]), float(full[o]))


@pytest.mark.parametrize('need_dgrad', [True, False])
def test_batched_weight_preparation_matches_the_per_layer_launches(ops, need_dgrad):
    """layers.WeightBatch (three launches per network: zero the slots, pfst_weight_prep_batched, pfst_conv_pack_weight_f16x2_batched) writes
    byte for byte the images and the maxima of the per-layer path (pfst_absmax / pfst_wino_filter_plain / pfst_conv_pack_weight_f16x2),
    for every convolution of the DeepLabV3+ segmentor, and follows the weights when they change."""
    from helpers import model_cfg
    from pfst_amd import layers
    from pfst_amd.registry import SEGMENTORS
    from pfst_amd.synthetic import fill_state_dict
    old_math, old_enabled = layers.CONV_MATH, layers.WeightBatch.enabled
    layers.CONV_MATH = 'f16x3'
    try:
        model = SEGMENTORS.build(model_cfg())
        fill_state_dict(model.state_dict(), 3)
        model.to(DEV)
        images = ('w4f', 'w4d', 'uf', 'ud', 'wf', 'wd', 'w6f', 'w6d')
        maxima = ('w_amax', 'uf_amax', 'ud_amax')

        def snapshot():
            out = {}
            for name, c in model.named_modules():
                if not isinstance(c, layers.Conv2dP):
                    continue
                for a in images:
                    if getattr(c, a) is not None:
                        out[name, a] = getattr(c, a).clone()
                for a in maxima:
                    if getattr(c, a) is not None:
                        out[name, a] = getattr(c, a).view(-1, ops.AMAX_SUB).max(dim=1).values.clone()
                out[name, 'modes'] = (c.wino, c.wino_f16, c.f16_f, c.f16_d, c.split_f, c.split_d)
            return out
        for round_ in range(2):
            layers.WeightBatch.enabled = True
            model.repack_weights(need_dgrad)
            got = snapshot()
            layers.WeightBatch.enabled = False
            for c in model.convs():          # forget the batch's buffers: the per-layer path allocates its own maxima
                c.w_amax = c.uf_amax = c.ud_amax = None
            model.repack_weights(need_dgrad)
            want = snapshot()
            assert got.keys() == want.keys()
            n_f16 = 0
            for k in want:
                if k[1] == 'modes':
                    assert got[k] == want[k], k
                    n_f16 += want[k][1] or want[k][2] or want[k][3]
                else:
                    assert torch.equal(got[k], want[k]), k
            assert n_f16 > 40               # the batch really covered the network
            with torch.no_grad():           # next round: other weights, same tables
                for p in model.parameters():
                    p.mul_(1.7).add_(0.01)
    finally:
        layers.CONV_MATH, layers.WeightBatch.enabled = old_math, old_enabled


@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('case', [(3, 64, 256, 24, 1), (2, 128, 96, 20, 1), (2, 64, 64, 16, 3), (1, 192, 130, 33, 1), (8, 256, 512, 32, 1)])
def test_f16x3_minmax_partials_predict_the_normalised_maximum(ops, case, relu):
    """pfst_conv_igemm_f16x3(stats_minmax) + pfst_bn_finalize_partials(minmax): the GEMM epilogue emits per-channel (minimum, maximum)
    partials of its output beside the (sum, sum of squares) ones, and the finalize kernel maps the channel extrema through the layer's own
    fma(x, sc, sh) [+ ReLU]: the slot group receives max |[relu](bn(x))| of a tensor that has not been (and need not be) written --
    bit for bit what pfst_bn_apply(y_amax) publishes when it writes it, with gammas of both signs, ragged row / pixel tiles, chained tiles
    (8 images of 1024 pixels) and the 64-row tile.  The sums and every output value are unchanged by the extra partials."""
    n, ci, co, hw, k = case
    x = (torch.randn(n, ci, hw, hw, generator=g(ci)) * 2.0).to(DEV)
    w = (torch.randn(co, ci, k, k, generator=g(co)) * 0.1).to(DEV)
    gamma = (torch.randn(co, generator=g(3)) * 0.8).to(DEV)          # both signs: the maximum output may come from the minimum input
    gamma[0] = 0.0
    beta = (torch.randn(co, generator=g(4)) * 0.5).to(DEV)
    w4f, _, wa = ops.pack_weight_f16x2(w, True, False)
    xa = ops.absmax(x)
    pad = k // 2
    y0, st0, sl = ops.conv_fprop_f16x3(x, w4f, wa, xa, co, k, 1, 1, pad, want_stats=True)
    sums0 = st0[:2 * co * sl].clone()
    y, st, sl1 = ops.conv_fprop_f16x3(x, w4f, wa, xa, co, k, 1, 1, pad, want_stats=True, want_minmax=True)
    assert sl1 == sl and torch.equal(y, y0) and torch.equal(st[:2 * co * sl], sums0)
    mm = st[2 * co * sl:4 * co * sl].view(co, sl, 2)
    lo, hi = mm[:, :, 0].min(dim=1)[0], mm[:, :, 1].max(dim=1)[0]
    assert torch.equal(lo, y.amin(dim=(0, 2, 3))) and torch.equal(hi, y.amax(dim=(0, 2, 3)))
    slots = ops.amax_slots(x.device)
    mean, invstd, coef = ops.bn_finalize_partials(st, sl, co, n * hw * hw, gamma=gamma, beta=beta, predict_amax=slots, relu=relu)
    true = ops.amax_slots(x.device)
    yn = ops.bn_apply(y, mean, invstd, gamma, beta, relu, amax=true)
    assert float(slots.max()) == float(true.max()) == float(yn.abs().max()), (float(slots.max()), float(true.max()), float(yn.abs().max()))


@pytest.mark.parametrize('case', [(2, 256, 512, 16, 16), (1, 64, 256, 32, 16), (3, 80, 256, 16, 8), (2, 96, 512, 8, 16), (8, 512, 2048, 32, 32),
                                  (2, 128, 256, 12, 16), (2, 2048, 512, 16, 16), (2, 560, 512, 16, 24)])
def test_f16x3_gemm_and_wgrad_normalise_on_load(ops, case):
    """pfst_conv_igemm_f16x3(bnl) / pfst_conv_wgrad_f16x3(bnl): the 1x1 convolution after a conv -> BN -> ReLU layer reads that layer's
    PRE-normalisation output and applies max(fma(x, sc, sh), 0) between load and split (Bottleneck conv2 -> bn2 -> relu -> conv3,
    /root/reference/rsiseg/models/backbones/resnet.py:282-290): the same fma and max per element as pfst_bn_apply, the same scale (the
    maximum of the normalised tensor), so output, fused statistics and -- in deterministic mode -- the weight gradient are BIT-IDENTICAL to the
    launches on the written tensor.  Cases: even / odd step counts (tile chains or not), a last half channel block (80), gammas of both signs,
    eight images of chained tiles, ragged pixel tiles (12 x 16: the tail columns hold normalised zeros and must stay out of the statistics)."""
    n, ci, co, H, W = case
    pre = (torch.randn(n, ci, H, W, generator=g(ci)) * 1.5).to(DEV)
    gamma = (torch.randn(ci, generator=g(3)) * 0.8).to(DEV)
    gamma[0] = 0.0
    beta = (torch.randn(ci, generator=g(4)) * 0.5).to(DEV)
    w = (torch.randn(co, ci, 1, 1, generator=g(co)) * 0.1).to(DEV)
    dy = torch.randn(n, co, H, W, generator=g(7)).to(DEV)
    assert ops.conv_fprop_bnl_ok(ci, co, 1)
    mean, invstd, coef = ops.bn_stats(pre, gamma=gamma, beta=beta)
    ya = ops.amax_slots(pre.device)
    y = ops.bn_apply(pre, mean, invstd, gamma, beta, True, amax=ya)
    w4f, _, wa = ops.pack_weight_f16x2(w, True, False)
    ref, st_ref, sl = ops.conv_fprop_f16x3(y, w4f, wa, ya, co, 1, want_stats=True, want_minmax=True)
    st_ref = st_ref[:4 * co * sl].clone()
    out, st, sl1 = ops.conv_fprop_f16x3(pre, w4f, wa, ya, co, 1, want_stats=True, want_minmax=True, bnl=coef)
    assert sl1 == sl and torch.equal(out, ref), float((out - ref).abs().max())
    assert torch.equal(st[:4 * co * sl], st_ref)
    assert torch.equal(ops.conv_fprop_f16x3(pre, w4f, wa, ya, co, 1, bnl=coef), ref)                 # without the statistics epilogue
    da = ops.absmax(dy)
    ops.set_deterministic(True)
    try:
        dw_ref = ops.conv_wgrad_f16x3_(torch.zeros(co, ci, 1, 1, device=DEV), y, dy, ya, da)
        dw = ops.conv_wgrad_f16x3_(torch.zeros(co, ci, 1, 1, device=DEV), pre, dy, ya, da, bnl=coef)
    finally:
        ops.set_deterministic(False)
    assert torch.equal(dw, dw_ref), float((dw - dw_ref).abs().max())
    dw2 = ops.conv_wgrad_f16x3_(torch.zeros(co, ci, 1, 1, device=DEV), pre, dy, ya, da, bnl=coef)   # the default (atomic) mode: to summation order
    assert_close(dw2, dw_ref.double(), 1e-6, 'wgrad bnl')


@pytest.mark.parametrize('case', [(2, 128, 128, 16, 16, 1), (1, 256, 192, 32, 24, 2), (8, 128, 256, 32, 32, 1)])
def test_wino_output_emits_minmax_partials(ops, case):
    """pfst_wino_output(stats_minmax): the Winograd output transform writes per-channel (minimum, maximum) partials of its output behind the
    (sum, sum of squares) ones -- what pfst_bn_finalize_partials needs to predict max |relu(bn(y))| for a consumer that normalises on load
    (Bottleneck conv2 through the Winograd domain -> bn2 -> conv3).  Output and sums unchanged, extrema exact."""
    n, ci, co, H, W, d = case
    x = torch.randn(n, ci, H, W, generator=g(1)).to(DEV)
    w = (torch.randn(co, ci, 3, 3, generator=g(2)) * 0.1).to(DEV)
    uf, _ = ops.wino_pack_weight(w, True, False)
    y0, st0, sl = ops.wino_conv(x, uf, co, d, want_stats=True)
    sums0 = st0[:2 * co * sl].clone()
    y, st, sl1 = ops.wino_conv(x, uf, co, d, want_stats=True, want_minmax=True)
    assert sl1 == sl and torch.equal(y, y0) and torch.equal(st[:2 * co * sl], sums0)
    mm = st[2 * co * sl:4 * co * sl].view(co, sl, 2)
    assert torch.equal(mm[:, :, 0].min(dim=1)[0], y.amin(dim=(0, 2, 3))) and torch.equal(mm[:, :, 1].max(dim=1)[0], y.amax(dim=(0, 2, 3)))
    gamma = (torch.randn(co, generator=g(3)) * 0.8).to(DEV)
    beta = (torch.randn(co, generator=g(4)) * 0.5).to(DEV)
    slots, true = ops.amax_slots(x.device), ops.amax_slots(x.device)
    mean, invstd, coef = ops.bn_finalize_partials(st, sl, co, n * H * W, gamma=gamma, beta=beta, predict_amax=slots, relu=True)
    yn = ops.bn_apply(y, mean, invstd, gamma, beta, True, amax=true)
    assert float(slots.max()) == float(true.max()) == float(yn.abs().max())


@pytest.mark.parametrize('case', [(2, 64, 40, 48, 1), (2, 40, 32, 32, 4), (1, 24, 33, 37, 2), (2, 16, 256, 256, 1)])
def test_depthwise_kernels_emit_minmax_partials(ops, case):
    """pfst_dwconv3x3(stats_minmax) / pfst_dwconv3x3_multi_fwd(stats_minmax): the depthwise kernels (strips, whole planes, the three-branch
    launch) write per-channel (minimum, maximum) partials of their outputs behind the sums, for the pointwise layer that normalises on load
    (layers.FOLD_BN_DWSEP).  Outputs and sums unchanged, extrema exact, predicted max |relu(bn(y))| = what bn_apply publishes."""
    n, c, H, W, d = case
    x = torch.randn(n, c, H, W, generator=g(1)).to(DEV)
    w = torch.randn(c, 1, 3, 3, generator=g(2)).to(DEV)
    gamma = (torch.randn(c, generator=g(3)) * 0.8).to(DEV)
    beta = (torch.randn(c, generator=g(4)) * 0.5).to(DEV)

    def check(y, st, sl, y0, sums0):
        assert torch.equal(y, y0) and torch.equal(st[:2 * c * sl], sums0)
        mm = st[2 * c * sl:4 * c * sl].view(c, sl, 2)
        assert torch.equal(mm[:, :, 0].min(dim=1)[0], y.amin(dim=(0, 2, 3))) and torch.equal(mm[:, :, 1].max(dim=1)[0], y.amax(dim=(0, 2, 3)))
        slots, true = ops.amax_slots(x.device), ops.amax_slots(x.device)
        mean, invstd, coef = ops.bn_finalize_partials(st, sl, c, n * H * W, gamma=gamma, beta=beta, predict_amax=slots, relu=True)
        yn = ops.bn_apply(y, mean, invstd, gamma, beta, True, amax=true)
        assert float(slots.max()) == float(true.max()) == float(yn.abs().max())

    y0, st0, sl = ops.dwconv(x, w, d, want_stats=True)
    sums0 = st0[:2 * c * sl].clone()
    y, st, sl1 = ops.dwconv(x, w, d, want_stats=True, want_minmax=True)
    assert sl1 == sl
    check(y, st, sl, y0, sums0)
    dils = [4 * d, 8 * d]
    if ops.dwconv_multi_ok(x, dils):
        ws = [w, torch.randn(c, 1, 3, 3, generator=g(5)).to(DEV)]
        r0 = ops.dwconv_multi(x, ws, dils, want_stats=True)
        keep = [(a.clone(), b[:2 * c * k].clone()) for a, b, k in r0]
        r1 = ops.dwconv_multi(x, ws, dils, want_stats=True, want_minmax=True)
        for (yy, ss, kk), (ya, sa) in zip(r1, keep):
            check(yy, ss, kk, ya, sa)


def test_depthwise_backward_is_exact_beside_a_weight_gradient_kernel(ops):
    """Round 5: the fused depthwise backward with the BatchNorm-backward fold (dwconv3x3_kernel<3, true>) returned wrong sums for one of its
    nine weight-gradient accumulators, on a few channels, whenever the f16x3 weight-gradient kernel of another stream shared the CUs --
    the schedule the product's stream overlap creates -- in the build whose streaming kernels contained packed-fp32 (v_pk_*) code; exact
    without it (pfst_amd/build.py NO_SLP, tools/race_probe.py).  Eight launches beside a busy side stream must reproduce the quiet launch bit
    for bit (one image per weight-gradient slot here: two workgroups per channel, their fp32 atomics commute exactly only if ... they do
    not: the comparison runs in deterministic mode)."""
    n, c, h, w = 2, 560, 32, 32
    x = torch.randn(n, c, h, w, generator=g(1)).to(DEV)
    dy = torch.randn(n, c, h, w, generator=g(2)).to(DEV)
    wt = torch.randn(c, 1, 3, 3, generator=g(3)).to(DEV)
    gamma = (torch.rand(c, generator=g(4)) + 0.5).to(DEV)
    beta = (torch.randn(c, generator=g(5)) * 0.1).to(DEV)
    pre = ops.dwconv(x, wt, 1)
    mean, invstd, coef = ops.bn_stats(pre, gamma=gamma, beta=beta)
    X = torch.randn(2, 512, 64, 64, generator=g(6)).to(DEV)
    DY = torch.randn(2, 512, 64, 64, generator=g(7)).to(DEV)
    DW = torch.zeros(512 * 512, device=DEV)
    xa, dya = ops.absmax(X), ops.absmax(DY)
    side = torch.cuda.Stream()
    ops.set_deterministic(True)
    try:
        rec = ops.bn_backward_sums(dy, pre, mean, invstd, gamma, beta, torch.zeros(c, device=DEV), torch.zeros(c, device=DEV))
        for bnl in (None, coef):
            ref = torch.zeros(c * 9, device=DEV)
            dxr = torch.empty_like(x)
            ops.dwconv_bwd_(ref, x, dy, wt, 1, dxr, bnl=bnl, bnb=(pre, rec))
            torch.cuda.synchronize()
            for rep in range(8):
                dw = torch.zeros(c * 9, device=DEV)
                dx = torch.empty_like(x)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(4):
                        ops.conv_wgrad_f16x3_(DW, X, DY, xa, dya)
                ops.dwconv_bwd_(dw, x, dy, wt, 1, dx, bnl=bnl, bnb=(pre, rec))
                torch.cuda.synchronize()
                assert torch.equal(dx, dxr)
                bad = (dw != ref).nonzero().flatten().tolist()
                assert not bad, f'run {rep}: weight-gradient elements {bad[:8]} (taps {[i % 9 for i in bad[:8]]}) differ from the quiet launch'
    finally:
        ops.set_deterministic(False)


@pytest.mark.parametrize('slots', [2, 5])
def test_f16x3_tile_chain_is_the_same_arithmetic(ops, slots):
    """The f16x3 GEMMs walk up to 8 tiles per workgroup as one software pipeline (the next tile's first pairs are loaded, split and stored
    inside the current tile's last steps).  With the grid sized for a handful of slots (pfst_f16x3_set_slots) small problems take that
    path -- chains of 2 ... 8 tiles, tiles of several images / filter sets in one chain, ragged last tiles, the fused statistics and the
    fused BatchNorm-backward sums, accumulation -- and must reproduce the one-tile-per-workgroup launch bit for bit."""
    torch.manual_seed(5)

    def run():
        out = {}
        # 1x1 fprop with statistics: K = 64 (2 steps: the load side runs two tiles ahead), 128, 560 (a half-empty last block), ragged rows / pixels
        for ci, co, hw in ((64, 256, 24), (128, 96, 20), (560, 128, 16), (192, 130, 33)):
            x = torch.randn(3, ci, hw, hw, generator=g(ci)).to(DEV) * 3.0
            w = (torch.randn(co, ci, 1, 1, generator=g(co)) * 0.1).to(DEV)
            w4f, w4d, wa = ops.pack_weight_f16x2(w, True, ops.f16x3_eligible(co, ci, 1))
            y, st, sl = ops.conv_fprop_f16x3(x, w4f, wa, ops.absmax(x), co, 1, want_stats=True)
            out['y', ci, co] = y.clone()
            out['stats', ci, co] = st[:2 * co * sl].clone()
            if w4d is not None:
                dy = torch.randn(3, co, hw, hw, generator=g(1)).to(DEV) * 1e-3
                dx = ops.conv_dgrad_f16x3(dy, w4d, wa, ops.absmax(dy), ci, (hw, hw), 1)
                out['dx', ci, co] = dx.clone()
                out['dx2', ci, co] = ops.conv_dgrad_f16x3(dy, w4d, wa, ops.absmax(dy), ci, (hw, hw), 1, out=dx, accumulate=True).clone()
        # data gradient with the fused BatchNorm-backward sums (whole 128-row tiles), with and without the residual form
        n, co, ci, hw = 2, 96, 256, 32
        w = (torch.randn(co, ci, 1, 1, generator=g(3)) * 0.1).to(DEV)
        _, w4d, wa = ops.pack_weight_f16x2(w, False, True)
        dy = torch.randn(n, co, hw, hw, generator=g(4)).to(DEV)
        pre = torch.randn(n, ci, hw, hw, generator=g(5)).to(DEV)
        gamma, beta = torch.rand(ci, generator=g(6)).to(DEV) + 0.5, (torch.randn(ci, generator=g(7)) * 0.1).to(DEV)
        _, _, coef = ops.bn_stats(pre, gamma=gamma, beta=beta)
        yres = torch.relu(torch.randn(n, ci, hw, hw, generator=g(8))).to(DEV)
        for name, bnb in (('relu', (pre, None, coef, True)), ('residual', (pre, yres, coef, True)), ('plain', (pre, None, coef, False))):
            dx, part, sl = ops.conv_dgrad_f16x3(dy, w4d, wa, ops.absmax(dy), ci, (hw, hw), 1, bnb=bnb)
            out['bnb', name] = dx.clone()
            out['bnb_part', name] = part[:2 * ci * sl].clone()
        # the 36 transform-domain GEMMs of a Winograd layer (tiles of different filter sets in one chain) and their data gradient
        x = torch.randn(2, 128, 24, 24, generator=g(9)).to(DEV)
        w3 = (torch.randn(160, 128, 3, 3, generator=g(10)) * 0.1).to(DEV)
        uf, ud, af, ad = ops.wino_pack_weight_f16(w3)
        out['wino'] = ops.wino_conv(x, uf, 160, 2, u_amax=af).clone()
        out['wino_d'] = ops.wino_conv(out['wino'], ud, 128, 2, u_amax=ad).clone()
        # ... and with 256 output rows: the 256-row workgroup tile (512 threads), packed and plain activations
        w4 = (torch.randn(256, 128, 3, 3, generator=g(11)) * 0.1).to(DEV)
        uf4, _, af4, _ = ops.wino_pack_weight_f16(w4, True, False)
        out['wino256'] = ops.wino_conv(x, uf4, 256, 1, u_amax=af4).clone()
        for ci, co, hw in ((128, 512, 20), (192, 256, 33)):
            xx = torch.randn(2, ci, hw, hw, generator=g(ci + 1)).to(DEV)
            ww = (torch.randn(co, ci, 1, 1, generator=g(co + 1)) * 0.1).to(DEV)
            w4f, _, wa = ops.pack_weight_f16x2(ww, True, False)
            y, st, sl = ops.conv_fprop_f16x3(xx, w4f, wa, ops.absmax(xx), co, 1, want_stats=True)
            out['y256', ci, co] = y.clone()
            out['stats256', ci, co] = st[:2 * co * sl].clone()
        return out
    try:
        ops.set_f16x3_slots(slots)
        chained = run()
        ops.set_f16x3_slots(1 << 20)          # more slots than tiles: one workgroup per tile
        single = run()
    finally:
        ops.set_f16x3_slots(0)
    assert chained.keys() == single.keys()
    for k in single:
        assert torch.equal(chained[k], single[k]), k
