"""The f16x3 arithmetic's ELEMENT-WISE guarantee, stated and tested per direction.

Every operand tensor t of an f16x3 GEMM is scaled by s = 2^(14 - floor(log2 T)), T = the tensor's scale bound (its measured max |t|;
for the pre-split Winograd operands a norm bound of it), and split into two fp16 pieces h = fp16(t s), l = fp16(t s - h):

    |t - (h + l) / s| <= 2^-22 |t| + 2^-39 T        (h, l normal: 22 significand bits; below 2^-18 T the second piece is subnormal
                                                     or zero and only the absolute term is left)

A contraction sum_k a_k b_k evaluated as ah bh + ah bl + al bh (al bl <= 2^-22 |a b| dropped) in fp32 accumulators therefore obeys

    |err| <= 2^-19 sum_k |a_k b_k|  +  2^-38 K A B                                                          (*)

(3 x 2^-22 from the splits and the dropped term, the rest of the 2^-19 for the fp32 accumulation -- the share the fp32-input MFMA kernel
pays too; A, B = the two tensors' scale bounds).  So: fp32-faithful NORM-wise, and element-wise for every output whose products are not
all more than 2^-18 below the tensors' maxima; an output row fed ONLY by operand rows 2^-20 / 2^-30 below the maximum keeps the
absolute 2^-38 K A B -- about 1e-6 / 1e-3 of its own size -- which is inside north_star's mixed tolerance (1e-4 max + 1e-3 |ref|) and
irrelevant to training, but it is NOT fp32's relative accuracy and is not claimed to be.  bf16x6 (fp32's exponent range per piece) and
the fp32-input MFMA keep the relative term alone on the same rows; they are printed beside f16x3 for scale.

The tests build operands whose pixel columns / channel rows sit at 1, 2^-10, 2^-20 and 2^-30 of the tensor maximum and assert (*) for EVERY
output element against fp64, plus the pure relative bound 2^-19 sum |a b| on the classes at 1 and 2^-10."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'
REL, ABS = 2.0 ** -19, 2.0 ** -38


@pytest.fixture(scope='module')
def ops():
    from pfst_amd import hip_ops
    return hip_ops


def g(seed):
    return torch.Generator().manual_seed(seed)


def classes_along(n, block=8):
    """class index 0..3 per position: blocks of `block` positions at 2^0, 2^-10, 2^-20, 2^-30"""
    return (torch.arange(n) // block) % 4


def report(name, err, abs_sum, ref, cls, K, A, B, check_abs=True):
    """err, abs_sum, ref: fp64 tensors; cls: class index broadcastable to them.  Prints per class the worst |err| / sum|ab| (the relative
    accuracy of that class) and returns the worst ratio against the full bound (*)"""
    bound = REL * abs_sum + ABS * K * A * B
    worst = float((err / bound).max())
    rows = []
    for c in range(4):
        m = (cls == c).expand_as(err)
        if m.any():
            rows.append((c, float((err[m] / abs_sum[m].clamp_min(1e-300)).max()), float(err[m].max() / ref[m].abs().max().clamp_min(1e-300))))
    print(f'   {name:28s} worst/bound {worst:7.3f}   ' + '  '.join(f'2^-{10 * c}: rel-to-sum|ab| {r:.1e} rel-to-max {q:.1e}' for c, r, q in rows))
    return worst, {c: r for c, r, _ in rows}


@pytest.mark.parametrize('case', [(2, 256, 128, 16, 32, 1, 1), (2, 512, 256, 8, 32, 1, 1), (1, 128, 128, 24, 32, 3, 2)])
def test_fprop_and_dgrad_elementwise(ops, case):
    n, ci, co, H, W, k, d = case
    p = d * (k // 2)
    cls_w = classes_along(W)                                           # pixel columns in four magnitude classes
    x = torch.randn(n, ci, H, W, generator=g(1)) * 2.0 ** (-10.0 * cls_w)
    w = torch.randn(co, ci, k, k, generator=g(2)) * 0.05
    dy = torch.randn(n, co, H, W, generator=g(3)) * 1e-4 * 2.0 ** (-10.0 * cls_w)
    xd, wd, dyd = x.to(DEV), w.to(DEV), dy.to(DEV)
    ref = F.conv2d(x.double(), w.double(), None, 1, p, d)
    asum = F.conv2d(x.double().abs(), w.double().abs(), None, 1, p, d)
    dref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), 1, p, d)
    dasum = torch.nn.grad.conv2d_input(x.shape, w.double().abs(), dy.double().abs(), 1, p, d)
    A, Bw = float(x.abs().max()), float(w.abs().max())
    Ady = float(dy.abs().max())
    print(f'\ncase {case}:')
    w4f, w4d, wa = ops.pack_weight_f16x2(wd, True, True)
    w6f, w6d = ops.pack_weight_split(wd, True, True)
    wf, wdg = ops.pack_weight(wd, True, True)
    outs = {
        'f16x3 fprop': ops.conv_fprop_f16x3(xd, w4f, wa, ops.absmax(xd), co, k, 1, d, p),
        'bf16x6 fprop': ops.conv_fprop_split(xd, w6f, co, k, 1, d, p),
        'fp32-MFMA fprop': ops.conv_fprop(xd, wf, co, k, 1, d, p),
    }
    res = {}
    for name, y in outs.items():
        res[name] = report(name, (y.double().cpu() - ref).abs(), asum, ref, cls_w, ci * k * k, A, Bw)
    douts = {
        'f16x3 dgrad': ops.conv_dgrad_f16x3(dyd, w4d, wa, ops.absmax(dyd), ci, (H, W), k, 1, d, p),
        'bf16x6 dgrad': ops.conv_dgrad_split(dyd, w6d, ci, (H, W), k, 1, d, p),
        'fp32-MFMA dgrad': ops.conv_dgrad(dyd, wdg, ci, (H, W), k, 1, d, p),
    }
    for name, dx in douts.items():
        res[name] = report(name, (dx.double().cpu() - dref).abs(), dasum, dref, cls_w, co * k * k, Ady, Bw)
    for name in ('f16x3 fprop', 'f16x3 dgrad'):
        worst, by_class = res[name]
        assert worst <= 1.0, (name, worst)                                  # (*) on every element
        assert by_class[0] <= REL and by_class[1] <= REL, (name, by_class)  # down to 2^-10 (in fact 2^-18): fp32's relative accuracy
    # the other two arithmetics hold the relative term alone on every class (their pieces carry fp32's exponent range): the yardstick
    for name in ('bf16x6 fprop', 'bf16x6 dgrad', 'fp32-MFMA fprop', 'fp32-MFMA dgrad'):
        assert max(res[name][1].values()) <= REL, (name, res[name][1])
    if k == 1:
        # and the absolute term is REAL for f16x3: the 2^-30 class has lost its relative accuracy (otherwise the statement above is too weak)
        assert res['f16x3 fprop'][1][3] > 2.0 ** -22


@pytest.mark.parametrize('case', [(2, 256, 128, 16, 24), (3, 128, 512, 8, 32)])
def test_wgrad_1x1_elementwise(ops, case):
    """dw[co, ci] = sum over pixels of dy[co, p] x[ci, p]: channel ROWS of x (columns of dw) in the four magnitude classes, and once more
    with the rows of dy scaled instead"""
    n, ci, co, H, W = case
    for which in ('x rows', 'dy rows'):
        cx = classes_along(ci, 16) if which == 'x rows' else torch.zeros(ci, dtype=torch.long)
        cy = classes_along(co, 16) if which == 'dy rows' else torch.zeros(co, dtype=torch.long)
        x = torch.randn(n, ci, H, W, generator=g(1)) * 2.0 ** (-10.0 * cx.view(1, ci, 1, 1))
        dy = torch.randn(n, co, H, W, generator=g(3)) * 1e-4 * 2.0 ** (-10.0 * cy.view(1, co, 1, 1))
        xd, dyd = x.to(DEV), dy.to(DEV)
        ref = torch.einsum('nohw,nihw->oi', dy.double(), x.double())
        asum = torch.einsum('nohw,nihw->oi', dy.double().abs(), x.double().abs())
        cls = (cx.view(1, ci) + cy.view(co, 1))
        K, A, B = n * H * W, float(x.abs().max()), float(dy.abs().max())
        print(f'\ncase {case}, {which}:')
        res = {}
        dw = torch.zeros(co, ci, 1, 1, device=DEV)
        ops.conv_wgrad_f16x3_(dw, xd, dyd, ops.absmax(xd), ops.absmax(dyd))
        res['f16x3'] = report('f16x3 wgrad', (dw.double().cpu().view(co, ci) - ref).abs(), asum, ref, cls, K, A, B)
        dw6 = torch.zeros(co, ci, 1, 1, device=DEV)
        ops.conv_wgrad_split_(dw6, xd, dyd, 1)
        res['bf16x6'] = report('bf16x6 wgrad', (dw6.double().cpu().view(co, ci) - ref).abs(), asum, ref, cls, K, A, B)
        dw32 = torch.zeros(co, ci, 1, 1, device=DEV)
        ops.conv_wgrad_(dw32, xd, dyd, 1)
        res['f32'] = report('fp32-MFMA wgrad', (dw32.double().cpu().view(co, ci) - ref).abs(), asum, ref, cls, K, A, B)
        worst, by_class = res['f16x3']
        assert worst <= 1.0, worst
        assert by_class[0] <= REL and by_class[1] <= REL, by_class
        assert max(res['bf16x6'][1].values()) <= REL and max(res['f32'][1].values()) <= REL


@pytest.mark.parametrize('case', [(2, 128, 128, 32, 32, 1), (1, 256, 128, 32, 64, 2)])
def test_winograd_presplit_elementwise_and_what_the_norm_bound_costs(ops, case):
    """The Winograd layers' input transform writes V PRE-SPLIT, so its scale must be known before V exists: it comes from max |x| through the
    transform's norm bound (|B^T d B| <= 100 max |d| < 2^7), i.e. up to 7 bits above the true max |V| -- elements of V below 2^-11 (instead
    of 2^-18) of the true maximum fall back to the absolute term.  Measured here against the same kernels with the scale from the TRUE
    max |V| (plain V + in-register split: pfst_wino_input without pack_x_amax, v_packed = 0) and against the bf16x6 Winograd GEMM, on
    image columns at 2^0 / 2^-10 / 2^-20 / 2^-30 of the maximum.  F(4x4) in fp32 carries its own transform rounding (3e-5 norm-wise,
    tests/test_hip_ops.py::WINO_TOL), so the relative term of the bound here is that of the fp32 Winograd pipeline; the absolute term
    is (*)'s with the 2^7 of the norm bound: 2^-31 K A B."""
    from pfst_amd._lib import call
    n, ci, co, H, W, d = case
    m = 4
    cls_w = classes_along(W, W // 4)                                   # whole 4x4 output tiles (x d) per class
    x = torch.randn(n, ci, H, W, generator=g(1)) * 2.0 ** (-10.0 * cls_w)
    w = torch.randn(co, ci, 3, 3, generator=g(2)) * 0.05
    xd, wd = x.to(DEV), w.to(DEV)
    ref = F.conv2d(x.double(), w.double(), None, 1, d, d)
    asum = F.conv2d(x.double().abs(), w.double().abs(), None, 1, d, d)
    A, Bw, K = float(x.abs().max()), float(w.abs().max()), 9 * ci
    uf, _, af, _ = ops.wino_pack_weight_f16(wd, True, False, m=m)
    u6, _ = ops.wino_pack_weight_split(wd, True, False, m=m)
    u32, _ = ops.wino_pack_weight(wd, True, False, m=m)
    print(f'\ncase {case}:')

    def run_true_max():
        """the same pipeline with V written plain and scaled from its measured maximum"""
        nx = (m + 2) ** 2
        t = ops.wino_tiles(H, W, d, m)
        v = torch.empty(nx * n * ci * t, device=DEV)
        mb = torch.empty(nx * n * co * t, device=DEV)
        va = ops.amax_slots(xd.device)
        st = torch.cuda.current_stream().cuda_stream
        call('pfst_wino_input', xd.data_ptr(), ci * H * W, v.data_ptr(), n, ci, H, W, d, m, va.data_ptr(), 0, 0, st)
        call('pfst_wino_gemm_f16x3', v.data_ptr(), uf.data_ptr(), af.data_ptr(), va.data_ptr(), mb.data_ptr(), n, ci, co, t, m, 0, st)
        y = torch.empty(n, co, H, W, device=DEV)
        call('pfst_wino_output', mb.data_ptr(), y.data_ptr(), co * H * W, n, co, H, W, d, 0, 0, 0, 0, 0, 0, 0, m, st)
        return y, float(va.max())

    y_true, vmax = run_true_max()
    bound_used = 100.0 * A
    print(f'   max|x| {A:.3f}  true max|V| {vmax:.3f}  norm bound 100 max|x| {bound_used:.1f}: {torch.log2(torch.tensor(bound_used / vmax)):.1f} bits given away')
    outs = {'f16x3 pre-split (norm bound)': ops.wino_conv(xd, uf, co, d, m=m, u_amax=af),
            'f16x3 true max|V|': y_true,
            'bf16x6 Winograd': ops.wino_conv(xd, u6, co, d, m=m),
            'fp32-MFMA Winograd': ops.wino_conv(xd, u32, co, d, m=m)}
    WREL = 2.0 ** -13                  # fp32 F(4x4): worst element of the transform rounding, relative to sum |x||w| (measured ~2^-15)
    res = {}
    for name, y in outs.items():
        err = (y.double().cpu() - ref).abs()
        bound = WREL * asum + 2.0 ** -31 * K * A * Bw
        worst = float((err / bound).max())
        per = {c: float((err[..., cls_w == c] / asum[..., cls_w == c]).max()) for c in range(4)}
        res[name] = (worst, per)
        print(f'   {name:30s} worst/bound {worst:7.3f}   ' + '  '.join(f'2^-{10 * c}: {r:.1e}' for c, r in per.items()))
    for name in ('f16x3 pre-split (norm bound)', 'f16x3 true max|V|'):
        assert res[name][0] <= 1.0, (name, res[name])
        assert res[name][1][0] <= WREL and res[name][1][1] <= WREL, (name, res[name][1])
    assert max(res['bf16x6 Winograd'][1].values()) <= WREL


def test_split_instructions_against_the_definition(ops):
    """The pinned instruction sequences the GEMM loops issue (round 4: v_fma_mixlo_f16 / v_fma_mixhi_f16, four instructions per pair of values
    instead of six) and the plain code of the prologues / packing kernels give, bit for bit, the two pieces of the definition
        h = fp16_rne(x s),   l = fp16_rne(x s - h)          s = 2^(14 - floor(log2 max |x|))
    (NumPy float16 conversion = IEEE round-to-nearest-even with gradual underflow), over 24 binades below the maximum incl. values whose
    second -- or first -- piece is subnormal or zero, exact powers of two, ties, both signs and zeros."""
    import numpy as np
    from pfst_amd._lib import call
    rng = np.random.RandomState(0)
    n = 1 << 16
    mag = np.exp2(-rng.uniform(0, 30, n)).astype(np.float32)
    x = (rng.standard_normal(n).astype(np.float32) * mag)
    x[:64] = np.exp2(-(np.arange(64) % 31)).astype(np.float32) * np.where(np.arange(64) % 2 == 0, 1, -1)       # exact powers of two
    x[64:128] = (1.0 + 2.0 ** -11) * np.exp2(-(np.arange(64) % 24)).astype(np.float32)                        # ties of the first piece
    x[128:136] = 0.0
    x[136] = 3.999                                                                                                # the maximum: s x just below 2^15
    amax_val = float(np.abs(x).max())
    xd = torch.from_numpy(x).to(DEV)
    amax = ops.absmax(xd)
    outs = [torch.empty(n, dtype=torch.int32, device=DEV) for _ in range(3)]
    call('pfst_f16x3_split_probe', xd.data_ptr(), n, amax.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
         torch.cuda.current_stream().cuda_stream)
    loop8, loop4, plain = (o.cpu().numpy().view(np.uint32) for o in outs)
    s = np.float32(2.0 ** (14 - int(np.floor(np.log2(amax_val)))))
    xs = x * s                                                       # exact: a power of two, no overflow / underflow in fp32 here
    h = xs.astype(np.float16)
    l = (xs - h.astype(np.float32)).astype(np.float16)               # the remainder is exact in fp32
    want = h.view(np.uint16).astype(np.uint32) | (l.view(np.uint16).astype(np.uint32) << 16)
    # -0.0 second pieces: the sign of a zero remainder is not part of the definition (x s - h = +0 either way in the sum)
    canon = lambda a: np.where((a >> 16) == 0x8000, a & 0xffff, a)
    for name, got in (('8-value loop sequence', loop8), ('4-value loop sequence', loop4), ('plain code', plain)):
        bad = np.nonzero(canon(got) != canon(want))[0]
        assert bad.size == 0, (name, bad[:5], [hex(v) for v in got[bad[:5]]], [hex(v) for v in want[bad[:5]]], x[bad[:5]])
    assert np.isfinite(h.astype(np.float32)).all() and float(np.abs(h.astype(np.float32)).max()) < 2.0 ** 15
    lbits = want >> 16
    sub = ((lbits & 0x7c00) == 0) & ((lbits & 0x3ff) != 0)          # exponent field 0, mantissa != 0
    print(f'\n   {n} values: {int(sub.sum())} with a subnormal second piece, {int(((lbits & 0x7fff) == 0).sum())} with none, all three paths bit-identical to the definition')
    assert sub.sum() > 100
