"""Thin torch-tensor front end of the C ABI (include/pfst_hip.h).

Every function checks device / dtype / layout on the host (a wrong shape must never reach a kernel),
passes raw device pointers + the current HIP stream, and returns torch tensors it allocated.
There is NO fallback: a missing libpfst_hip.so or a CPU tensor raises.
Tensors may be channel slices of a bigger NCHW tensor (batch stride != C*H*W)."""
import ctypes
import os

import torch

from ._lib import call, lib

F32, U8, I64, F64 = torch.float32, torch.uint8, torch.int64, torch.float64


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype=F32, nd=None):
    if not t.is_cuda:
        raise RuntimeError('pfst_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback')
    if t.dtype != dtype:
        raise TypeError(f'expected {dtype}, got {t.dtype}')
    if nd is not None and t.dim() != nd:
        raise ValueError(f'expected {nd}-d tensor, got shape {tuple(t.shape)}')
    return t


def _bs(t):
    """batch stride of an NCHW tensor whose (C,H,W) part is dense."""
    _chk(t, F32, 4)
    n, c, h, w = t.shape
    if w > 1 and t.stride(3) != 1 or (h > 1 and t.stride(2) != w) or (c > 1 and t.stride(1) != h * w):
        raise ValueError(f'tensor is not a dense-plane NCHW view: shape {tuple(t.shape)} strides {t.stride()}')
    return t.stride(0) if n > 1 else max(t.stride(0), c * h * w)


def _dense(t, dtype=F32):
    _chk(t, dtype)
    if not t.is_contiguous():
        raise ValueError('tensor must be contiguous')
    return t


def _p(t):
    return 0 if t is None else t.data_ptr()


# ---------------------------------------------------------------- utilities
def set_deterministic(on=True):
    """pfst_set_deterministic: sums that are normally completed by atomic adds of several workgroups (split-K weight gradients, BatchNorm-backward
    reductions, depthwise weight gradients, bias gradients, PFGSTLoss source statistics) are formed in a fixed order -- per-workgroup (per grid
    slice) partials in a scratch + an ordered reduction; same launch shapes.  The gradient arena of a step is then bit-identical run to run and for
    any stream schedule (tests/test_deterministic_gpu.py); the b = 8 x 1024^2 step is about 3 % slower (tools/det_cost.py).  What the reference's `--deterministic`
    (cudnn.deterministic = True, rsiseg/apis/train.py:63-66) asks for."""
    call('pfst_set_deterministic', int(bool(on)))


def is_deterministic():
    return bool(lib().pfst_get_deterministic())


_h2d_stage = {}
_h2d_const = {}


def h2d_small(t_cpu, dev, tag):
    """a small per-step host tensor (augmentation parameters, class choices) to the device WITHOUT the stream synchronisation a pageable
    copy implies (`.to(dev)` of pageable memory makes the host wait for everything queued on the stream -- the host then starts the next
    pass with no lead over the device): staged through a pinned buffer cached per (tag, shape, dtype), copied asynchronously.  A buffer is
    reused one step later at the earliest; the step's blocking read of its log values lies in between"""
    key = (tag, tuple(t_cpu.shape), t_cpu.dtype)
    st = _h2d_stage.get(key)
    if st is None:
        st = _h2d_stage[key] = torch.empty(t_cpu.shape, dtype=t_cpu.dtype, pin_memory=True)
    st.copy_(t_cpu)
    return st.to(dev, non_blocking=True)


def const_tensor(values, dev, dtype=F32):
    """a device tensor of a few host constants (normalisation mean / std ...), built once per (values, device)"""
    key = (tuple(float(v) for v in values), str(dev), dtype)
    t = _h2d_const.get(key)
    if t is None:
        t = _h2d_const[key] = torch.tensor(list(values), dtype=dtype, device=dev)
    return t


def fill_(t, value):
    _dense(t)
    call('pfst_fill_f32', t.data_ptr(), t.numel(), float(value), _stream())
    return t


def axpy_(y, x, alpha=1.0):
    _dense(y), _dense(x)
    assert y.numel() == x.numel()
    call('pfst_axpy_f32', y.data_ptr(), x.data_ptr(), float(alpha), y.numel(), _stream())
    return y


def to_u8(lab):
    if lab.dtype == U8:
        return _dense(lab, U8)
    _dense(lab, I64)
    out = torch.empty(lab.shape, dtype=U8, device=lab.device)
    call('pfst_i64_to_u8', lab.data_ptr(), out.data_ptr(), lab.numel(), _stream())
    return out


def to_i64(lab):
    _dense(lab, U8)
    out = torch.empty(lab.shape, dtype=I64, device=lab.device)
    call('pfst_u8_to_i64', lab.data_ptr(), out.data_ptr(), lab.numel(), _stream())
    return out


# ---------------------------------------------------------------- dense conv
def conv_out_size(hi, k, stride, dil, pad):
    return (hi + 2 * pad - (k - 1) * dil - 1) // stride + 1


def pack_weight(w, want_fprop=True, want_dgrad=True, out_f=None, out_d=None):
    """w [Cout][Cin][k][k] -> K-major packings used by the implicit GEMM."""
    _dense(w)
    co, ci, kh, kw = w.shape
    assert kh == kw and kh in (1, 3)
    wf = (out_f if out_f is not None else torch.empty(kh * kw * ci, co, device=w.device)) if want_fprop else None
    wd = (out_d if out_d is not None else torch.empty(kh * kw * co, ci, device=w.device)) if want_dgrad else None
    call('pfst_conv_pack_weight', w.data_ptr(), _p(wf), _p(wd), co, ci, kh * kw, _stream())
    return wf, wd


_stats_cache = {}


def _stats_ws(dev, nfloat):
    """per-stream scratch for the conv epilogue's BN partials (consumed by bn_finalize_partials on the same stream)"""
    key = (dev, torch.cuda.current_stream().cuda_stream)
    t = _stats_cache.get(key)
    if t is None or t.numel() < nfloat:
        t = torch.empty(max(nfloat, 1 << 20), dtype=F32, device=dev)
        _stats_cache[key] = t
    return t


def _stats_ws_multi(dev, nfloat, i):
    """as _stats_ws, one scratch per branch of a multi-branch launch (all of them are alive until their statistics are finalised)"""
    key = (dev, torch.cuda.current_stream().cuda_stream, 'multi', i)
    t = _stats_cache.get(key)
    if t is None or t.numel() < nfloat:
        t = torch.empty(max(nfloat, 1 << 16), dtype=F32, device=dev)
        _stats_cache[key] = t
    return t


def conv_stats_slots(n, cout, ho, wo):
    return n * lib().pfst_conv_stats_slots(cout, ho, wo)


def bn_finalize_partials(stats, slots, c, count, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, gamma=None, beta=None,
                         predict_amax=None, relu=True, coef_out=None):
    """gamma / beta given: also returns coef [C,4] = (mean, invstd, sc, sh), the record the fused BatchNorm-backward sums read.
    predict_amax: a zeroed slot group (amax_slots); `stats` then carries the producer's (minimum, maximum) partials behind the sums
    (conv_fprop_f16x3(want_minmax=True)) and the group receives max |[relu](bn(x))| -- what bn_apply(amax=...) would publish, known before
    (or without) the normalised tensor being written"""
    mean = torch.empty(c, device=stats.device)
    invstd = torch.empty(c, device=stats.device)
    coef = torch.empty(c, 4, device=stats.device) if gamma is not None else None
    if coef_out is not None:                 # rows of a concat buffer's coefficient table (engine.Var.coef_table)
        assert gamma is not None and tuple(coef_out.shape) == (c, 4) and coef_out.is_contiguous()
        coef = coef_out
    mm = 0
    if predict_amax is not None:
        assert gamma is not None and stats.numel() >= 4 * c * slots
        mm = stats.data_ptr() + 4 * 2 * c * slots
    call('pfst_bn_finalize_partials', stats.data_ptr(), slots, c, float(count), mean.data_ptr(), invstd.data_ptr(),
         _p(running_mean), _p(running_var), momentum, eps, _p(gamma), _p(beta), _p(coef), mm, int(relu), _p(predict_amax), _stream())
    return (mean, invstd, coef) if gamma is not None else (mean, invstd)


class _BnbFuseStruct(ctypes.Structure):           # pfst_bnb_fuse_t (include/pfst_hip.h)
    _fields_ = [('x', ctypes.c_void_p), ('x_bs', ctypes.c_longlong), ('y', ctypes.c_void_p), ('y_bs', ctypes.c_longlong),
                ('coef', ctypes.c_void_p), ('partials', ctypes.c_void_p), ('relu', ctypes.c_int), ('y_mask', ctypes.c_void_p)]


def bnb_tile_rows(m):
    return 128 if m > 64 else (64 if m > 32 else 32)


def conv_fprop(x, wk, cout, ksize, stride=1, dil=1, pad=0, bias=None, out=None, want_stats=False):
    """want_stats: also return (stats_ws, slots), the epilogue's per-channel BN partial sums"""
    n, c, hi, wi = x.shape
    ho, wo = conv_out_size(hi, ksize, stride, dil, pad), conv_out_size(wi, ksize, stride, dil, pad)
    assert wk.numel() == ksize * ksize * c * cout, (wk.shape, c, cout, ksize)
    if out is None:
        out = torch.empty(n, cout, ho, wo, device=x.device)
    assert tuple(out.shape) == (n, cout, ho, wo)
    slots = conv_stats_slots(n, cout, ho, wo) if want_stats else 0
    st = _stats_ws(x.device, 2 * cout * slots) if want_stats else None
    call('pfst_conv_igemm', x.data_ptr(), _bs(x), _dense(wk).data_ptr(), _p(bias), out.data_ptr(), _bs(out),
         n, c, hi, wi, cout, ho, wo, ksize, stride, dil, pad, 0, 0, _p(st), 0, _stream())
    return (out, st, slots) if want_stats else out


def conv_dgrad(dy, wk_d, cin, in_hw, ksize, stride=1, dil=1, pad=0, out=None, accumulate=False, bnb=None):
    """bnb = (pre, y | None, coef, relu): `out` is the COMPLETE gradient of a conv -> BN layer's output and this launch also emits
    that layer's BatchNorm-backward sums (pfst_bnb_fuse_t); returns (out, partials, slots) then."""
    n, co, ho, wo = dy.shape
    hi, wi = in_hw
    assert wk_d.numel() == ksize * ksize * co * cin
    if out is None:
        assert not accumulate
        out = torch.empty(n, cin, hi, wi, device=dy.device)
    assert tuple(out.shape) == (n, cin, hi, wi)
    fuse, part, slots, st = 0, None, 0, None
    if bnb is not None:
        st, part, slots = _bnb_struct(bnb, n, cin, hi, wi, co, dy.device)
        fuse = ctypes.addressof(st)
    call('pfst_conv_igemm', dy.data_ptr(), _bs(dy), _dense(wk_d).data_ptr(), 0, out.data_ptr(), _bs(out),
         n, co, ho, wo, cin, hi, wi, ksize, stride, dil, pad, 1, int(accumulate), 0, fuse, _stream())
    return (out, part, slots) if bnb is not None else out


def pack_weight_split(w, want_fprop=True, want_dgrad=True, out_f=None, out_d=None):
    """w [Cout][Cin][k][k] -> bf16x3-split K-major images (uint8 buffers of 6 bytes per weight)."""
    _dense(w)
    co, ci, kh, kw = w.shape
    nbytes = 6 * co * ci * kh * kw
    wf = (out_f if out_f is not None else torch.empty(nbytes, dtype=U8, device=w.device)) if want_fprop else None
    wd = (out_d if out_d is not None else torch.empty(nbytes, dtype=U8, device=w.device)) if want_dgrad else None
    call('pfst_conv_pack_weight_split', w.data_ptr(), _p(wf), _p(wd), co, ci, kh * kw, _stream())
    return wf, wd


def conv_fprop_split(x, wk6, cout, ksize, stride=1, dil=1, pad=0, bias=None, out=None, want_stats=False):
    n, c, hi, wi = x.shape
    ho, wo = conv_out_size(hi, ksize, stride, dil, pad), conv_out_size(wi, ksize, stride, dil, pad)
    assert wk6.numel() == 6 * ksize * ksize * c * cout
    if out is None:
        out = torch.empty(n, cout, ho, wo, device=x.device)
    slots = conv_stats_slots(n, cout, ho, wo) if want_stats else 0
    st = _stats_ws(x.device, 2 * cout * slots) if want_stats else None
    call('pfst_conv_igemm_split', x.data_ptr(), _bs(x), wk6.data_ptr(), _p(bias), out.data_ptr(), _bs(out),
         n, c, hi, wi, cout, ho, wo, ksize, stride, dil, pad, 0, 0, _p(st), 0, _stream())
    return (out, st, slots) if want_stats else out


# ---------------------------------------------------------------- fp32-faithful two-piece fp16 split (csrc/conv_f16x3.hip)
AMAX_SUB = 1024        # floats per slot group (csrc/amax.h)
_amax_arena = {}


def amax_slots(dev, groups=1):
    """`groups` zeroed slot groups (groups x 1024 fp32) carved out of a zero-filled arena block: one memset per 4096 groups instead of one
    per tensor.  A block that is used up is replaced by a fresh one; the old one lives as long as views of it do."""
    need = groups * AMAX_SUB
    # one arena per (device, stream): a block is zero-filled on the stream that allocates it, and the first kernel to touch a group
    # (the producer publishing into it) runs on that same stream; consumers on other streams are ordered behind the producer anyway
    key = (dev, torch.cuda.current_stream().cuda_stream)
    blk = _amax_arena.get(key)
    if blk is None or blk[1] + need > blk[0].numel():
        blk = [torch.zeros(max(1 << 22, need), dtype=F32, device=dev), 0]
        _amax_arena[key] = blk
    out = blk[0][blk[1]:blk[1] + need]
    blk[1] += need
    return out


def absmax(x, planes=1, out=None):
    """max |x| -> device slot group(s) of 1024 fp32 (csrc/amax.h).  planes = 1: one group for the whole tensor (a dense tensor, or a dense-plane NCHW view such as a
    channel slice of a concat buffer); planes > 1: x is dense and viewed as `planes` equal contiguous parts, one slot each.
    out: slots to extend (already zeroed or holding an earlier maximum)"""
    if out is None:
        out = amax_slots(x.device, planes)
    if planes == 1 and x.dim() == 4 and not x.is_contiguous():
        n, c, h, w = x.shape
        call('pfst_absmax', x.data_ptr(), c * h * w, n, _bs(x), 0, out.data_ptr(), _stream())
        return out
    _dense(x)
    n = x.numel() // planes
    assert n * planes == x.numel()
    call('pfst_absmax', x.data_ptr(), n, planes, n, 1, out.data_ptr(), _stream())
    return out


def set_f16x3_slots(slots):
    """resident workgroup slots the f16x3 GEMMs size their tile-chain grid for (0: the device's two per CU) -- a test hook"""
    call('pfst_f16x3_set_slots', int(slots))


def f16x3_eligible(cin, cout, ksize=3):
    """the f16x3 implicit GEMM covers contractions over whole 32-channel blocks -- a 1x1 convolution also a last half block: the loads of
    the missing 16 channels fall outside their buffers' ranges and return zeros (the range check includes the scalar offset on gfx950:
    tools/probes/soffset_range_probe.hip) -- and more than 32 output rows (33 ... 64: the 64-row tile, one 32-row block per wave)"""
    return (cin % 32 == 0 or (ksize == 1 and cin % 16 == 0)) and cout > F16X3_MIN_ROWS


F16X3_MIN_ROWS = 32


def pack_weight_f16x2(w, want_fprop=True, want_dgrad=True, out_f=None, out_d=None, amax=None, sets=1):
    """w [Cout][Cin][k][k] (or `sets` plain [Cout][Cin] Winograd-domain filter sets) -> two-piece fp16 K-major images (uint8 buffers of
    4 bytes per weight per layout) + the slots with max |w| per set that fixed their scales"""
    _dense(w)
    if sets == 1:
        co, ci, kh, kw = w.shape
        t = kh * kw
    else:
        assert w.dim() == 3 and w.shape[0] == sets
        _, co, ci = w.shape
        t = 1
    nbytes = 4 * sets * co * ci * t
    wf = (out_f if out_f is not None else torch.empty(nbytes, dtype=U8, device=w.device)) if want_fprop else None
    wd = (out_d if out_d is not None else torch.empty(nbytes, dtype=U8, device=w.device)) if want_dgrad else None
    if amax is None:
        amax = absmax(w, sets)
    call('pfst_conv_pack_weight_f16x2', w.data_ptr(), _p(wf), _p(wd), co, ci, t, sets, amax.data_ptr(), _stream())
    return wf, wd, amax


class _WeightJob(ctypes.Structure):             # pfst_weight_job_t (include/pfst_hip.h)
    _fields_ = [('src', ctypes.c_void_p), ('dst_f', ctypes.c_void_p), ('dst_d', ctypes.c_void_p), ('amax_f', ctypes.c_void_p),
                ('amax_d', ctypes.c_void_p), ('Cout', ctypes.c_int), ('Cin', ctypes.c_int), ('T', ctypes.c_int), ('sets', ctypes.c_int),
                ('m', ctypes.c_int), ('first_block', ctypes.c_int)]


class WeightJobTable:
    """a job table of the batched weight preparation: the host copy the library checks and its device copy the kernel reads.
    jobs: dicts with the fields of pfst_weight_job_t (tensors or None for the pointers); pack: 0 = prep table, 1 = pack table"""

    def __init__(self, jobs, pack, dev):
        self.n = len(jobs)
        self.host = (_WeightJob * self.n)()
        first = 0
        for j, d in zip(self.host, jobs):
            j.src, j.dst_f, j.dst_d, j.amax_f, j.amax_d = (_p(d.get(k)) or None for k in ('src', 'dst_f', 'dst_d', 'amax_f', 'amax_d'))
            j.Cout, j.Cin, j.T, j.sets, j.m, j.first_block = d['Cout'], d['Cin'], d['T'], d.get('sets', 1), d.get('m', 0), first
            first += lib().pfst_weight_job_blocks(ctypes.addressof(j), pack)
        self.blocks = first
        self.dev = torch.frombuffer(bytearray(bytes(self.host)), dtype=U8).to(dev)
        self.keep = [t for d in jobs for t in d.values() if torch.is_tensor(t)]     # the buffers the table points into

    def run(self, name):
        call(name, ctypes.addressof(self.host), self.dev.data_ptr(), self.n, _stream())


BNL_MAX_C = 2048          # coefficient rows the normalising GEMM keeps in LDS (conv_f16x3.hip BNL_ROWS)


def conv_fprop_bnl_ok(cin, cout, ksize, stride=1, pad=0):
    """can conv_fprop_f16x3 normalise its input as it loads (bnl=...): the 256-row pixel-to-pixel tile, coefficient rows in LDS"""
    return ksize == 1 and stride == 1 and pad == 0 and cout % 256 == 0 and cin <= BNL_MAX_C and f16x3_eligible(cin, cout, 1)


def conv_fprop_f16x3(x, wk4, w_amax, x_amax, cout, ksize, stride=1, dil=1, pad=0, bias=None, out=None, want_stats=False, want_minmax=False,
                     bnl=None):
    """want_minmax (with want_stats): the statistics scratch also receives the per-channel (minimum, maximum) partials of the output behind
    the sums -- bn_finalize_partials(predict_amax=...) turns them into max |relu(bn(out))|.
    bnl: coef [C, 4] of the conv -> BN -> ReLU layer feeding this one: x is that layer's PRE-normalisation output, normalised between load and
    split (x_amax = the predicted max of the normalised tensor)"""
    n, c, hi, wi = x.shape
    assert bnl is None or (tuple(bnl.shape) == (c, 4) and bias is None and conv_fprop_bnl_ok(c, cout, ksize, stride, pad))
    ho, wo = conv_out_size(hi, ksize, stride, dil, pad), conv_out_size(wi, ksize, stride, dil, pad)
    assert wk4.numel() == 4 * ksize * ksize * c * cout and f16x3_eligible(c, cout, ksize)
    if out is None:
        out = torch.empty(n, cout, ho, wo, device=x.device)
    slots = n * ((ho * wo + 127) // 128) * 2 if want_stats else 0        # two pixel-waves per 128-pixel tile at every tile height
    want_minmax = bool(want_minmax and want_stats and bias is None)
    st = _stats_ws(x.device, (4 if want_minmax else 2) * cout * slots) if want_stats else None
    call('pfst_conv_igemm_f16x3', x.data_ptr(), _bs(x), wk4.data_ptr(), w_amax.data_ptr(), x_amax.data_ptr(), _p(bias), out.data_ptr(), _bs(out),
         n, c, hi, wi, cout, ho, wo, ksize, stride, dil, pad, 0, 0, _p(st), 0, 0, 0, 0, int(want_minmax), _p(None if bnl is None else _dense(bnl)),
         _stream())
    return (out, st, slots) if want_stats else out


def conv_wgrad_f16x3_(dw, x, dy, x_amax, dy_amax, bnl=None):
    """dw += dL/dw of a stride-1 1x1 convolution with the f16x3 split (fp32 atomics); bnl: as conv_fprop_f16x3 (x is the pre-normalisation
    tensor, x_amax the predicted maximum of the normalised one)"""
    n, ci, h, w = x.shape
    co = dy.shape[1]
    assert dy.shape == (n, co, h, w) and dw.numel() == co * ci and (bnl is None or tuple(bnl.shape) == (ci, 4))
    call('pfst_conv_wgrad_f16x3', x.data_ptr(), _bs(x), dy.data_ptr(), _bs(dy), _dense(dw).data_ptr(), n, ci, co, h * w,
         x_amax.data_ptr(), dy_amax.data_ptr(), _p(None if bnl is None else _dense(bnl)), _stream())
    return dw


def dgrad_gate_ok(cin, in_hw):
    """can conv_dgrad_f16x3 add a ReLU-gated tensor in its epilogue (gate=...): whole 128-row tiles, whole 256-element mask groups"""
    return cin % 128 == 0 and (in_hw[0] * in_hw[1]) % 256 == 0


def conv_dgrad_f16x3(dy, wk4_d, w_amax, dy_amax, cin, in_hw, ksize, stride=1, dil=1, pad=0, out=None, accumulate=False, bnb=None, gate=None):
    """bnb: as conv_dgrad (returns (out, partials, slots) then);
    gate = (g, mask): out = data gradient + (mask bit ? g : 0) -- g an [N, cin, H, W] tensor, mask the ReLU bitmask bn_apply(want_mask=True)
    returned for a tensor of that shape (the identity branch of a residual block; `out` has no earlier writer: accumulate = False)"""
    n, co, ho, wo = dy.shape
    hi, wi = in_hw
    g_ptr, g_bs, m_ptr = 0, 0, 0
    if gate is not None:
        g, mask = gate
        assert not accumulate and dgrad_gate_ok(cin, in_hw) and tuple(g.shape) == (n, cin, hi, wi)
        assert mask.dtype == torch.int64 and mask.numel() == n * cin * hi * wi // 64
        g_ptr, g_bs, m_ptr = g.data_ptr(), _bs(g), mask.data_ptr()
    assert wk4_d.numel() == 4 * ksize * ksize * co * cin and f16x3_eligible(co, cin, ksize)
    if out is None:
        assert not accumulate
        out = torch.empty(n, cin, hi, wi, device=dy.device)
    fuse, part, slots, st = 0, None, 0, None
    if bnb is not None:
        st, part, slots = _bnb_struct(bnb, n, cin, hi, wi, co, dy.device)
        fuse = ctypes.addressof(st)
    call('pfst_conv_igemm_f16x3', dy.data_ptr(), _bs(dy), wk4_d.data_ptr(), w_amax.data_ptr(), dy_amax.data_ptr(), 0, out.data_ptr(), _bs(out),
         n, co, ho, wo, cin, hi, wi, ksize, stride, dil, pad, 1, int(accumulate), 0, fuse, g_ptr, g_bs, m_ptr, 0, 0, _stream())
    return (out, part, slots) if bnb is not None else out


def _bnb_struct(bnb, n, cin, hi, wi, co, dev):
    """-> (ctypes struct kept alive by the caller, partials tensor, slots) for a fused BatchNorm-backward data-gradient launch"""
    pre, y, coef, relu = bnb[:4]
    mask = bnb[4] if len(bnb) > 4 else None            # bn_apply's ReLU bitmask of y: the f16x3 epilogue reads the gate bits instead of y
    assert tuple(pre.shape) == (n, cin, hi, wi) and (y is None or tuple(y.shape) == tuple(pre.shape))
    assert mask is None or (y is not None and relu and (hi * wi) % 256 == 0 and mask.dtype == torch.int64 and mask.numel() == n * cin * hi * wi // 64)
    assert co % 16 == 0 and cin % bnb_tile_rows(cin) == 0 and tuple(coef.shape) == (cin, 4)
    slots = conv_stats_slots(n, cin, hi, wi)
    part = torch.empty(2 * cin * slots, dtype=F32, device=dev)     # owned by the consumer layer's context until its bn_backward ran
    st = _BnbFuseStruct(pre.data_ptr(), _bs(pre), _p(y), 0 if y is None else _bs(y), _dense(coef).data_ptr(), part.data_ptr(), int(relu), _p(mask))
    return st, part, slots


def conv_dgrad_split(dy, wk6_d, cin, in_hw, ksize, stride=1, dil=1, pad=0, out=None, accumulate=False, bnb=None):
    """bnb: as conv_dgrad (returns (out, partials, slots) then)"""
    n, co, ho, wo = dy.shape
    hi, wi = in_hw
    assert wk6_d.numel() == 6 * ksize * ksize * co * cin
    if out is None:
        assert not accumulate
        out = torch.empty(n, cin, hi, wi, device=dy.device)
    fuse, part, slots, st = 0, None, 0, None
    if bnb is not None:
        st, part, slots = _bnb_struct(bnb, n, cin, hi, wi, co, dy.device)
        fuse = ctypes.addressof(st)
    call('pfst_conv_igemm_split', dy.data_ptr(), _bs(dy), wk6_d.data_ptr(), 0, out.data_ptr(), _bs(out),
         n, co, ho, wo, cin, hi, wi, ksize, stride, dil, pad, 1, int(accumulate), 0, fuse, _stream())
    return (out, part, slots) if bnb is not None else out


def conv_wgrad_(dw, x, dy, ksize, stride=1, dil=1, pad=0):
    """dw += dL/dw (fp32 atomics)."""
    n, ci, hi, wi = x.shape
    _, co, ho, wo = dy.shape
    assert dy.shape[0] == n and dw.numel() == co * ci * ksize * ksize
    call('pfst_conv_wgrad', x.data_ptr(), _bs(x), dy.data_ptr(), _bs(dy), _dense(dw).data_ptr(), n, ci, hi, wi, co, ho, wo,
         ksize, stride, dil, pad, _stream())
    return dw


def conv_wgrad_split_(dw, x, dy, ksize, stride=1, dil=1, pad=0):
    """dw += dL/dw with the fp32-faithful bf16x6 split (fp32 atomics)."""
    n, ci, hi, wi = x.shape
    _, co, ho, wo = dy.shape
    assert dy.shape[0] == n and dw.numel() == co * ci * ksize * ksize
    call('pfst_conv_wgrad_split', x.data_ptr(), _bs(x), dy.data_ptr(), _bs(dy), _dense(dw).data_ptr(), n, ci, hi, wi, co, ho, wo,
         ksize, stride, dil, pad, _stream())
    return dw


def wgrad_q_operands_ok(x, dy):
    """the operand layout the K-quad weight-gradient kernels need beyond the layer's shape (pfst_wgrad_q_eligible, csrc/conv_wgrad_q.hip):
    16-byte aligned planes, batch strides in whole float4s.  A caller that fails this takes the generic kernel instead of an error."""
    return x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0 and _bs(x) % 4 == 0 and _bs(dy) % 4 == 0


def conv_wgrad_f16q_(dw, x, dy, x_amax, dy_amax, ksize, dil=1):
    """dw += dL/dw of a stride-1 'same' convolution (1x1, or 3x3 with pad == dil) with the f16x3 split on the K-quad kernel (fp32 atomics):
    the direct 3x3 layers of the stems / layer1 and the 1x1 layers with <= 64 output channels"""
    n, ci, h, w = x.shape
    co = dy.shape[1]
    assert tuple(dy.shape) == (n, co, h, w) and dw.numel() == co * ci * ksize * ksize
    call('pfst_conv_wgrad_f16x3_q', x.data_ptr(), _bs(x), dy.data_ptr(), _bs(dy), _dense(dw).data_ptr(), n, ci, h, w, co, ksize, dil,
         x_amax.data_ptr(), dy_amax.data_ptr(), _stream())
    return dw


def bias_grad_(db, dy):
    n, c, h, w = dy.shape
    call('pfst_bias_grad', dy.data_ptr(), _bs(dy), _dense(db).data_ptr(), n, c, h * w, _stream())
    return db


# ---------------------------------------------------------------- Winograd F(2x2,3x3) (wide stride-1 3x3 convolutions)
_wino_cache = {}


def _wino_ws(dev, tag, nfloat):
    """per-stream scratch for the transform-domain tensors (consumed on the same stream)"""
    key = (dev, torch.cuda.current_stream().cuda_stream, tag)
    t = _wino_cache.get(key)
    if t is None or t.numel() < nfloat:
        t = torch.empty(nfloat, dtype=F32, device=dev)
        _wino_cache[key] = t
    return t


# output tile edge m of the Winograd F(m x m, 3x3) transforms: 4 (4x fewer MACs, 36 transform indices) or 2 (2.25x, 16 indices)
WINO_TILE = 4


def _wino_m(m):
    m = WINO_TILE if m is None else int(m)
    assert m in (2, 4)
    return m, (m + 2) * (m + 2)


def set_wgrad_lds_pad(nbytes):
    """occupancy cap of the K-quad weight-gradient kernels (pfst_conv_wgrad_set_lds_pad)"""
    call('pfst_conv_wgrad_set_lds_pad', int(nbytes))


def wino_tiles(h, w, dil, m=None):
    return lib().pfst_wino_tiles(h, w, dil, _wino_m(m)[0])


def wino_pack_weight(w, want_fprop=True, want_dgrad=True, out_f=None, out_d=None, m=None):
    """w [Cout][Cin][3][3] -> transform-domain filters U[X][K/4][M][4], X = (m+2)^2, for fprop (K = Cin) and dgrad (K = Cout, flipped)."""
    _dense(w)
    m, nx = _wino_m(m)
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    uf = (out_f if out_f is not None else torch.empty(nx * co * ci, device=w.device)) if want_fprop else None
    ud = (out_d if out_d is not None else torch.empty(nx * co * ci, device=w.device)) if want_dgrad else None
    assert (uf is None or uf.numel() == nx * co * ci) and (ud is None or ud.numel() == nx * co * ci)
    call('pfst_wino_pack_weight', w.data_ptr(), _p(uf), _p(ud), co, ci, m, _stream())
    return uf, ud


def wino_pack_weight_split(w, want_fprop=True, want_dgrad=True, out_f=None, out_d=None, m=None):
    """transform-domain filters for the bf16x6 GEMM: X split-packed sets (uint8 buffers of X * 6 * Cout * Cin bytes)"""
    _dense(w)
    m, nx = _wino_m(m)
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    n = co * ci
    pf = _wino_ws(w.device, 'Pf', nx * n) if want_fprop else None
    pd = _wino_ws(w.device, 'Pd', nx * n) if want_dgrad else None
    call('pfst_wino_filter_plain', w.data_ptr(), _p(pf), _p(pd), co, ci, m, _stream())
    uf = (out_f if out_f is not None else torch.empty(nx * 6 * n, dtype=U8, device=w.device)) if want_fprop else None
    ud = (out_d if out_d is not None else torch.empty(nx * 6 * n, dtype=U8, device=w.device)) if want_dgrad else None
    assert (uf is None or uf.numel() == nx * 6 * n) and (ud is None or ud.numel() == nx * 6 * n)
    call('pfst_wino_pack_weight_split', _p(pf), _p(pd), _p(uf), _p(ud), co, ci, m, _stream())
    return uf, ud


def wino_pack_weight_f16(w, want_fprop=True, want_dgrad=True, out_f=None, out_d=None, m=None):
    """transform-domain filters for the f16x3 GEMM: X two-piece fp16 sets (uint8 buffers of X * 4 * Cout * Cin bytes) + the X slots with
    each set's absolute maximum; -> (uf, ud, amax_f, amax_d)"""
    _dense(w)
    m, nx = _wino_m(m)
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    n = co * ci
    pf = _wino_ws(w.device, 'Pf', nx * n) if want_fprop else None
    pd = _wino_ws(w.device, 'Pd', nx * n) if want_dgrad else None
    call('pfst_wino_filter_plain', w.data_ptr(), _p(pf), _p(pd), co, ci, m, _stream())
    uf = ud = af = ad = None
    if want_fprop:
        uf, _, af = pack_weight_f16x2(pf[:nx * n].view(nx, co, ci), True, False, out_f=out_f, sets=nx)
    if want_dgrad:
        _, ud, ad = pack_weight_f16x2(pd[:nx * n].view(nx, co, ci), False, True, out_d=out_d, sets=nx)
    return uf, ud, af, ad


def wino_conv(x, u, cout, dil, out=None, accumulate=False, keep_v=False, want_stats=False, m=None, u_amax=None, x_amax=None, bnl=None,
              want_minmax=False, bnb=None):
    """'same' 3x3 stride-1 convolution (or its data gradient, with the dgrad filter) through the transform domain.
    keep_v: the transformed input goes to a tensor of its own and is returned as (out, V) for the weight gradient
    (288 GB of HBM: keeping it resident beats re-transforming the input in backward).
    bnl: coef [C, 4] of the conv -> BN -> ReLU layer feeding this one: x is that layer's PRE-normalisation output, normalised by the input
    transform as it loads (x_amax then = the predicted max of the normalised tensor, bn_finalize_partials(predict_amax=...))"""
    n, c, h, w = x.shape
    assert bnl is None or (tuple(bnl.shape) == (c, 4) and (u_amax is None or x_amax is not None))
    m, nx = _wino_m(m)
    t = wino_tiles(h, w, dil, m)
    assert u.numel() == nx * c * cout * ((4 if u_amax is not None else 6) if u.dtype == U8 else 1), 'filter set was packed for another tile size'
    v = torch.empty(nx * n * c * t, dtype=F32, device=x.device) if keep_v else _wino_ws(x.device, 'V', nx * n * c * t)
    mb = _wino_ws(x.device, 'M', nx * n * cout * t)
    if out is None:
        assert not accumulate
        out = torch.empty(n, cout, h, w, device=x.device)
    assert tuple(out.shape) == (n, cout, h, w)
    # f16x3 (u_amax given): the input transform writes V PRE-SPLIT -- two fp16 pieces per element, scaled from max |x| (x_amax: the slot
    # group the producer of x published, else computed here) through the transform's norm bound -- and v_amax receives that bound
    v_amax = amax_slots(x.device) if u_amax is not None else None
    if u_amax is not None and x_amax is None:
        x_amax = absmax(x)
    call('pfst_wino_input', x.data_ptr(), _bs(x), v.data_ptr(), n, c, h, w, dil, m, _p(v_amax), _p(x_amax) if u_amax is not None else 0,
         _p(None if bnl is None else _dense(bnl)), _stream())
    if u_amax is not None:              # two-piece fp16 filter sets -> f16x3 GEMM
        call('pfst_wino_gemm_f16x3', v.data_ptr(), u.data_ptr(), u_amax.data_ptr(), v_amax.data_ptr(), mb.data_ptr(), n, c, cout, t, m, 1, _stream())
    else:
        gemm = 'pfst_wino_gemm_split' if u.dtype == U8 else 'pfst_wino_gemm'        # split-packed filters -> bf16x6 GEMM
        call(gemm, v.data_ptr(), _dense(u, u.dtype).data_ptr(), mb.data_ptr(), n, c, cout, t, m, _stream())
    slots, st = 0, None
    if want_stats:                      # BN partial sums of the output come out of the output transform
        slots = n * lib().pfst_wino_stats_slots(h, w, dil, m)
        st = _stats_ws(x.device, (4 if want_minmax else 2) * cout * slots)
    bx, bx_bs, bcoef, brelu = 0, 0, 0, 0
    if bnb is not None:
        # a data-gradient launch that completes the gradient of a conv -> BN [-> ReLU] layer's output (no residual): bnb = (pre, coef, relu) of
        # that layer -> returns (out, partials, slots) with its BatchNorm-backward sums for bn_backward(partials=...), no reduction pass
        assert not want_stats
        pre, coef, relu = bnb
        assert tuple(pre.shape) == (n, cout, h, w) and tuple(coef.shape) == (cout, 4)
        slots = n * lib().pfst_wino_stats_slots(h, w, dil, m)
        st = torch.empty(2 * cout * slots, dtype=F32, device=x.device)        # owned by the layer's context until its bn_backward ran
        bx, bx_bs, bcoef, brelu = pre.data_ptr(), _bs(pre), _dense(coef).data_ptr(), int(relu)
    call('pfst_wino_output', mb.data_ptr(), out.data_ptr(), _bs(out), n, cout, h, w, dil, int(accumulate), _p(st),
         int(bool(want_minmax and want_stats)), bx, bx_bs, bcoef, brelu, m, _stream())
    res = (out, st, slots) if (want_stats or bnb is not None) else (out,)
    if keep_v:
        res = res + ((v, v_amax),)          # the transformed input and the slot group with its absolute maximum (None unless f16x3)
    return res if len(res) > 1 else res[0]


def wino_wgrad_(dw, x, dy, dil, v=None, m=None, split=False, v_amax=None, x_amax=None, dy_amax=None):
    """dw += dL/dw of the 'same' 3x3 stride-1 convolution; v: the transformed input kept from the forward pass (same m);
    split: the transform-domain products with the fp32-faithful bf16x6 split instead of the fp32-input MFMA"""
    n, ci, h, w = x.shape
    co = dy.shape[1]
    m, nx = _wino_m(m)
    assert dy.shape == (n, co, h, w) and dw.numel() == co * ci * 9
    t = wino_tiles(h, w, dil, m)
    dm = _wino_ws(x.device, 'M', nx * n * co * t)
    du = _wino_ws(x.device, 'U', nx * co * ci)
    # split: False / 0 fp32-input MFMA, True / 1 bf16x6, 2 f16x3 (falls back to bf16x6 for <= 64 rows).  f16x3: both operands PRE-SPLIT by
    # their transforms (v given: the packed V and its bound group kept from the f16x3 forward pass)
    f16 = split == 2 and co > 64
    assert not (f16 and v is not None and v_amax is None), 'a kept V must come with the slot group of its scale bound'
    assert f16 or v_amax is None, 'a packed V kept from an f16x3 forward pass needs the f16x3 weight gradient'
    if v is None:
        v = _wino_ws(x.device, 'V', nx * n * ci * t)
        v_amax = amax_slots(x.device) if f16 else None
        if f16 and x_amax is None:
            x_amax = absmax(x)
        call('pfst_wino_input', x.data_ptr(), _bs(x), v.data_ptr(), n, ci, h, w, dil, m, _p(v_amax), _p(x_amax) if f16 else 0, 0, _stream())
    assert v.numel() >= nx * n * ci * t
    dm_amax = amax_slots(x.device) if f16 else None
    if f16 and dy_amax is None:
        dy_amax = absmax(dy)
    call('pfst_wino_dy', dy.data_ptr(), _bs(dy), dm.data_ptr(), n, co, h, w, dil, m, _p(dm_amax), _p(dy_amax) if f16 else 0, _stream())
    call('pfst_wino_wgrad', v.data_ptr(), dm.data_ptr(), du.data_ptr(), _dense(dw).data_ptr(), n, ci, co, t, m,
         2 if f16 else int(bool(split)), _p(v_amax), _p(dm_amax), int(f16), _stream())
    return dw


# ---------------------------------------------------------------- depthwise
def dwconv(x, w, dil, flip=False, out=None, accumulate=False, want_stats=False, bnl=None, want_minmax=False):
    """want_stats: also return (stats_ws, slots), per-channel BN partial sums of the output (as conv_fprop); want_minmax: followed by the
    (minimum, maximum) partials (as conv_fprop_f16x3);
    bnl: coef [C, 4] of the conv -> BN -> ReLU layer feeding this one: x is that layer's PRE-normalisation output, normalised on load"""
    n, c, h, wd = x.shape
    assert w.numel() == c * 9
    if out is None:
        assert not accumulate
        out = torch.empty(n, c, h, wd, device=x.device)
    slots, st = 0, None
    if want_stats:
        slots = n * lib().pfst_dwconv_stats_slots(h, wd, dil)
        st = _stats_ws(x.device, (4 if want_minmax else 2) * c * slots)
    call('pfst_dwconv3x3', x.data_ptr(), _bs(x), _dense(w).data_ptr(), out.data_ptr(), _bs(out), n, c, h, wd, dil,
         int(flip), int(accumulate), _p(st), int(bool(want_minmax and want_stats)), _p(bnl), _stream())
    return (out, st, slots) if want_stats else out


def dwconv_wgrad_(dw, x, dy, dil):
    n, c, h, w = x.shape
    assert dy.shape == x.shape and dw.numel() == c * 9
    call('pfst_dwconv3x3_wgrad', x.data_ptr(), _bs(x), dy.data_ptr(), _bs(dy), _dense(dw).data_ptr(), n, c, h, w, dil, _stream())
    return dw


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[0 if t is None else t.data_ptr() for t in tensors])


def dwconv_multi_ok(x, dils):
    """the fused multi-branch depthwise kernels cover this input (whole planes <= 16384 elements, W % 4 == 0, dilations % 4 == 0, dense
    16-byte aligned planes)"""
    n, c, h, w = x.shape
    if not (x.stride(3) == 1 and x.stride(2) == w and x.stride(1) == h * w):
        return False
    return (1 <= len(dils) <= 3 and _bs(x) % 4 == 0 and x.data_ptr() % 16 == 0
            and bool(lib().pfst_dwconv3x3_multi_ok(h, w, len(dils), (ctypes.c_int * len(dils))(*dils))))


def dwconv_multi(x, ws, dils, want_stats=False, want_mean=False, want_minmax=False):
    """y_i = depthwise 3x3 conv of x with filters ws[i] at dilation dils[i], every input plane staged ONCE for all branches
    -> [(y_i, stats_i, slots_i)] (stats as dwconv(want_stats=True): per-channel BN partials, one slot per image);
    want_mean: -> (that list, [N, C, 1, 1] plane means of x = global_avgpool(x)) from the same pass"""
    n, c, h, w = x.shape
    k = len(ws)
    ys = [torch.empty(n, c, h, w, device=x.device) for _ in range(k)]
    want_minmax = bool(want_minmax and want_stats)          # the (minimum, maximum) partials behind each branch's sums
    sts = [(_stats_ws_multi(x.device, (4 if want_minmax else 2) * c * n, i) if want_stats else None) for i in range(k)]
    mean = torch.empty(n, c, 1, 1, device=x.device) if want_mean else None
    call('pfst_dwconv3x3_multi_fwd', x.data_ptr(), _bs(x), k, _ptr_array([_dense(t) for t in ws]), _ptr_array(ys),
         (ctypes.c_longlong * k)(*[_bs(y) for y in ys]), _ptr_array(sts), int(want_minmax), (ctypes.c_int * k)(*dils), _p(mean), n, c, h, w,
         _stream())
    res = [(ys[i], sts[i], n if want_stats else 0) for i in range(k)]
    return (res, mean) if want_mean else res


def dwconv_multi_bwd_(dws, x, dys, ws, dils, dx, accumulate=False, mean_grad=None, bnb=None):
    """dws[i] += weight gradients, dx (+)= sum_i mirrored stencil of dys[i]: x read once, every dy once, dx written once;
    mean_grad: [N, C] gradient of the plane means dwconv_multi(want_mean=True) returned: dx += mean_grad / (H W);
    bnb = [(pre_i, rec_i)]: dys[i] are the gradients of the branches' BatchNorm + ReLU outputs (see dwconv_bwd_)"""
    n, c, h, w = x.shape
    k = len(ws)
    assert all(tuple(d.shape) == tuple(x.shape) and _bs(d) % 4 == 0 and d.data_ptr() % 16 == 0 for d in dys) and tuple(dx.shape) == tuple(x.shape)
    assert bnb is None or all(_bs(b[0]) == _bs(d) and tuple(b[0].shape) == tuple(d.shape) for b, d in zip(bnb, dys))
    call('pfst_dwconv3x3_multi_bwd', x.data_ptr(), _bs(x), k, _ptr_array([_dense(t) for t in ws]), _ptr_array(dys),
         (ctypes.c_longlong * k)(*[_bs(d) for d in dys]), _ptr_array([_dense(t) for t in dws]), (ctypes.c_int * k)(*dils),
         _p(None if mean_grad is None else _dense(mean_grad)), dx.data_ptr(), _bs(dx), int(accumulate),
         None if bnb is None else _ptr_array([b[0] for b in bnb]), None if bnb is None else _ptr_array([b[1] for b in bnb]),
         n, c, h, w, _stream())
    return dx


def dwconv_bwd_(dw, x, dy, w, dil, dx, accumulate=False, bnl=None, bnb=None):
    """both gradients of the depthwise convolution in one pass: dx (+)= the mirrored stencil of dy, dw += the weight gradient.
    bnb = (pre, rec): dy is the gradient of the layer's BatchNorm + ReLU OUTPUT, pre the convolution's own output, rec from
    bn_backward_sums: the gradient of the convolution output is formed while the rows are staged"""
    n, c, h, wd = x.shape
    assert dy.shape == x.shape and tuple(dx.shape) == tuple(x.shape) and dw.numel() == c * 9 and w.numel() == c * 9
    call('pfst_dwconv3x3_bwd', dy.data_ptr(), _bs(dy), x.data_ptr(), _bs(x), _dense(w).data_ptr(), dx.data_ptr(), _bs(dx),
         _dense(dw).data_ptr(), n, c, h, wd, dil, int(accumulate), _p(bnl), _p(None if bnb is None else bnb[0]),
         0 if bnb is None else _bs(bnb[0]), _p(None if bnb is None else bnb[1]), _stream())
    return dx


# ---------------------------------------------------------------- batch norm
_ws_cache = {}


def _ws(dev, nbytes=2 * 8 * 4096):
    key = (dev, torch.cuda.current_stream().cuda_stream)
    t = _ws_cache.get(key)
    if t is None or t.numel() * 8 < nbytes:
        t = torch.empty(max(nbytes // 8, 8192), dtype=F64, device=dev)
        _ws_cache[key] = t
    return t


def bn_stats(x, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, gamma=None, beta=None):
    n, c, h, w = x.shape
    mean = torch.empty(c, device=x.device)
    invstd = torch.empty(c, device=x.device)
    coef = torch.empty(c, 4, device=x.device) if gamma is not None else None
    call('pfst_bn_stats', x.data_ptr(), _bs(x), n, c, h * w, mean.data_ptr(), invstd.data_ptr(), _p(running_mean),
         _p(running_var), float(momentum), float(eps), _ws(x.device, 16 * c).data_ptr(), _p(gamma), _p(beta), _p(coef), _stream())
    return (mean, invstd, coef) if gamma is not None else (mean, invstd)


def bn_apply(x, mean, invstd, gamma, beta, relu=True, residual=None, out=None, want_mask=False, amax=None, post=None, residual_coef=None):
    """want_mask: also return the ReLU gate as a bitmask (int64 words) for bn_backward, or None where the kernel cannot
    produce it (plane size not a multiple of 256, unaligned slices) -- the caller then keeps using y.
    post: [N, C] factors applied after the ReLU (the Dropout2d mask of the layer feeding conv_seg, folded into this pass)"""
    n, c, h, w = x.shape
    assert post is None or (residual is None and post.numel() == n * c)
    assert residual_coef is None or (residual is not None and tuple(residual_coef.shape) == (c, 4))      # the residual is normalised on load
    if out is None:
        out = torch.empty(n, c, h, w, device=x.device)
    assert out.shape == x.shape
    mask = None
    if want_mask and relu and (h * w) % 256 == 0 and all(
            t is None or (_bs(t) % 4 == 0 and t.data_ptr() % 16 == 0) for t in (x, out, residual)):
        mask = torch.empty(n * c * h * w // 64, dtype=torch.int64, device=x.device)
    call('pfst_bn_apply', x.data_ptr(), _bs(x), _p(residual), 0 if residual is None else _bs(residual), out.data_ptr(), _bs(out),
         mean.data_ptr(), invstd.data_ptr(), _dense(gamma).data_ptr(), _dense(beta).data_ptr(), n, c, h * w, int(relu), _p(mask),
         _p(amax), _p(None if post is None else _dense(post)), _p(None if residual_coef is None else _dense(residual_coef)), _stream())
    return (out, mask) if want_mask else out


def bn_backward(dy, y, x, mean, invstd, gamma, dgamma, dbeta, relu=True, dres=None, dres_accumulate=False, dx=None, beta=None,
                mask=None, partials=None, slots=0, amax=None, post=None):
    """mask: the bitmask bn_apply(..., want_mask=True) returned; replaces y as the source of the ReLU gate.
    partials / slots: the (sum dz, sum dz*x) partials the launch that wrote dy emitted (conv_dgrad(bnb=...)): no reduction pass"""
    n, c, h, w = x.shape
    assert dy.shape == x.shape
    assert mask is None or (mask.dtype == torch.int64 and mask.numel() == n * c * h * w // 64 and (h * w) % 256 == 0)
    if dx is None:
        dx = torch.empty(n, c, h, w, device=x.device)
    call('pfst_bn_backward', dy.data_ptr(), _bs(dy), _p(y), 0 if y is None else _bs(y), x.data_ptr(), _bs(x),
         mean.data_ptr(), invstd.data_ptr(), _dense(gamma).data_ptr(), _p(beta), dx.data_ptr(), _bs(dx),
         _p(dres), 0 if dres is None else _bs(dres), int(dres_accumulate), _p(dgamma), _p(dbeta),
         n, c, h * w, int(relu), _p(mask), _ws(x.device, 16 * c).data_ptr(), _p(partials), int(slots), _p(amax), _p(None if post is None else _dense(post)), _stream())
    return dx


def bn_backward_dual(dy, mask, a, b):
    """the BatchNorm backward of TWO layers that receive the same gated gradient g = (mask bit ? dy : 0) -- bn3 and the downsample branch's BN of
    a stage's first Bottleneck -- in one reduction and one apply pass (pfst_bn_backward_dual).  a / b: dicts x (pre-BN tensor), mean, invstd,
    gamma, dgamma, dbeta, amax (slot group or None); a may carry partials / slots (its sums from the launch that wrote dy).  Returns
    (dxa, dxb), or None when the entry point does not take the case (deterministic mode, unaligned planes): the caller runs the layers one by one"""
    n, c, h, w = a['x'].shape
    assert dy.shape == a['x'].shape == b['x'].shape
    assert mask.dtype == torch.int64 and mask.numel() == n * c * h * w // 64 and (h * w) % 256 == 0
    if is_deterministic() or any(_bs(t) % 4 or t.data_ptr() % 16 for t in (dy, a["x"], b["x"])):
        return None
    dxa, dxb = torch.empty(n, c, h, w, device=dy.device), torch.empty(n, c, h, w, device=dy.device)
    ws = torch.empty(4 * c, dtype=torch.float64, device=dy.device)
    call('pfst_bn_backward_dual', dy.data_ptr(), _bs(dy), mask.data_ptr(),
         a['x'].data_ptr(), _bs(a['x']), a['mean'].data_ptr(), a['invstd'].data_ptr(), _dense(a['gamma']).data_ptr(), dxa.data_ptr(), _bs(dxa),
         _p(a['dgamma']), _p(a['dbeta']), ws.data_ptr(), _p(a.get('partials')), int(a.get('slots', 0)), _p(a.get('amax')),
         b['x'].data_ptr(), _bs(b['x']), b['mean'].data_ptr(), b['invstd'].data_ptr(), _dense(b['gamma']).data_ptr(), dxb.data_ptr(), _bs(dxb),
         _p(b['dgamma']), _p(b['dbeta']), ws[2 * c:].data_ptr(), _p(b.get('amax')), n, c, h * w, _stream())
    return dxa, dxb


def relu_gate_(out, g, mask, accumulate=False):
    """out (+)= g where bn_apply's ReLU bitmask `mask` has the element's bit"""
    n, c, h, w = g.shape
    assert tuple(out.shape) == tuple(g.shape) and mask.dtype == torch.int64 and mask.numel() == n * c * h * w // 64 and (h * w) % 256 == 0
    call('pfst_relu_gate', g.data_ptr(), _bs(g), mask.data_ptr(), out.data_ptr(), _bs(out), n, c, h * w, int(accumulate), _stream())
    return out


BN_BWD_REC_BYTES = 40          # sizeof(pfst_bn_bwd_rec_t): three doubles + four floats


def bn_backward_sums(dy, x, mean, invstd, gamma, beta, dgamma, dbeta, partials=None, slots=0):
    """the first half of bn_backward for a depthwise conv -> BN -> ReLU layer: the two sums (from `partials` or by the reduction pass),
    dgamma / dbeta += ..., and the per-channel record (uint8 [C * 40]) with which the layer's fused depthwise backward applies the second
    half itself while it stages dy and x (dwconv_bwd_(bnb=...), dwconv_multi_bwd_(bnb=...)): dL/dpre is never written"""
    n, c, h, w = x.shape
    assert dy.shape == x.shape
    rec = torch.empty(c * BN_BWD_REC_BYTES, dtype=U8, device=x.device)
    ws = torch.empty(2 * c, dtype=torch.float64, device=x.device)        # its own scratch: the record is read later, by another launch
    call('pfst_bn_backward_sums', dy.data_ptr(), _bs(dy), x.data_ptr(), _bs(x), mean.data_ptr(), invstd.data_ptr(),
         _dense(gamma).data_ptr(), _dense(beta).data_ptr(), _p(dgamma), _p(dbeta), n, c, h * w, ws.data_ptr(), _p(partials), int(slots),
         rec.data_ptr(), _stream())
    return rec


# ---------------------------------------------------------------- pooling / resize
def maxpool(x, bnl=None, amax=None):
    _dense(x)
    n, c, h, w = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty(n, c, ho, wo, device=x.device)
    idx = torch.empty(n, c, ho, wo, dtype=U8, device=x.device)
    call('pfst_maxpool3x3s2', x.data_ptr(), y.data_ptr(), idx.data_ptr(), n * c, h, w, ho, wo, _p(bnl), c, _p(amax), _stream())
    return y, idx


def maxpool_bwd(dy, idx, in_hw):
    _dense(dy)
    n, c, ho, wo = dy.shape
    dx = torch.empty(n, c, in_hw[0], in_hw[1], device=dy.device)
    call('pfst_maxpool3x3s2_bwd', dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), n * c, in_hw[0], in_hw[1], ho, wo, _stream())
    return dx


def resize_bilinear(x, size, out=None):
    n, c, hi, wi = x.shape
    if out is None:
        out = torch.empty(n, c, size[0], size[1], device=x.device)
    call('pfst_resize_bilinear', x.data_ptr(), _bs(x), out.data_ptr(), _bs(out), n, c, hi, wi, size[0], size[1], _stream())
    return out


def resize_bilinear_bwd(dy, in_hw, out=None, accumulate=False):
    n, c, ho, wo = dy.shape
    if out is None:
        assert not accumulate
        out = torch.empty(n, c, in_hw[0], in_hw[1], device=dy.device)
    call('pfst_resize_bilinear_bwd', dy.data_ptr(), _bs(dy), out.data_ptr(), _bs(out), n, c, in_hw[0], in_hw[1], ho, wo,
         int(accumulate), _stream())
    return out


def global_avgpool(x):
    n, c, h, w = x.shape
    y = torch.empty(n, c, 1, 1, device=x.device)
    call('pfst_global_avgpool', x.data_ptr(), _bs(x), y.data_ptr(), n, c, h * w, _stream())
    return y


def reduce_hw(dy):
    n, c, h, w = dy.shape
    v = torch.empty(n, c, 1, 1, device=dy.device)
    call('pfst_reduce_hw', dy.data_ptr(), _bs(dy), v.data_ptr(), n, c, h * w, _stream())
    return v


def broadcast_hw(v, out, scale=1.0, accumulate=False, amax=None):
    n, c, h, w = out.shape
    assert v.numel() == n * c
    call('pfst_broadcast_hw', _dense(v).data_ptr(), out.data_ptr(), _bs(out), n, c, h * w, float(scale), int(accumulate), _p(amax), _stream())
    return out


def upsample_nearest(x, factor):
    _dense(x)
    n, c, h, w = x.shape
    y = torch.empty(n, c, h * factor, w * factor, device=x.device)
    call('pfst_upsample_nearest', x.data_ptr(), y.data_ptr(), n * c, h, w, factor, _stream())
    return y


def upsample_nearest_bwd(dy, factor):
    _dense(dy)
    n, c, H, W = dy.shape
    dx = torch.empty(n, c, H // factor, W // factor, device=dy.device)
    call('pfst_upsample_nearest_bwd', dy.data_ptr(), dx.data_ptr(), n * c, H // factor, W // factor, factor, _stream())
    return dx


def channel_scale(x, mask):
    _dense(x)
    n, c, h, w = x.shape
    assert mask.numel() == n * c
    y = torch.empty_like(x)
    call('pfst_channel_scale', x.data_ptr(), _dense(mask).data_ptr(), y.data_ptr(), n, c, h * w, _stream())
    return y


# ---------------------------------------------------------------- test-time inference (sliding windows, flips, averaging)
def softmax_nchw(x, out=None):
    """F.softmax(x, dim=1) of an NCHW tensor (dense planes)"""
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty(n, c, h, w, device=x.device)
    call('pfst_softmax_nchw', x.data_ptr(), _bs(x), out.data_ptr(), _bs(out), n, c, h * w, _stream())
    return out


def window_accumulate_(preds, count, crop, y1, x1):
    """preds[:, :, y1:y1+hc, x1:x1+wc] += crop; count[:, :, y1:.., x1:..] += 1 (encoder_decoder.py:246-250)"""
    _dense(preds), _dense(count), _dense(crop)
    n, c, h, w = preds.shape
    hc, wc = crop.shape[-2:]
    assert crop.shape[:2] == (n, c) and count.numel() == n * h * w
    call('pfst_window_accumulate', crop.data_ptr(), preds.data_ptr(), count.data_ptr(), n, c, hc, wc, h, w, int(y1), int(x1), _stream())


def window_normalize_(preds, count):
    _dense(preds), _dense(count)
    n, c, h, w = preds.shape
    call('pfst_window_normalize', preds.data_ptr(), count.data_ptr(), n, c, h * w, _stream())
    return preds


def argmax_nchw(x):
    """x.argmax(dim=1) as uint8 [N, H, W] (first maximal class)"""
    n, c, h, w = x.shape
    lab = torch.empty(n, h, w, dtype=U8, device=x.device)
    call('pfst_argmax_nchw', x.data_ptr(), _bs(x), lab.data_ptr(), n, c, h * w, _stream())
    return lab


def flip_planes(x, horizontal=False, vertical=False):
    """x.flip(dims=(3,)) / (2,) of a dense NCHW tensor, out of place"""
    _dense(x)
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    call('pfst_flip_planes', x.data_ptr(), y.data_ptr(), n * c, h, w, int(horizontal), int(vertical), _stream())
    return y


def div_scalar_(x, d):
    _dense(x)
    call('pfst_div_scalar', x.data_ptr(), x.numel(), float(d), _stream())
    return x


# ---------------------------------------------------------------- losses
def ce_upsample_fwd(logits, label_u8, pix_weight=None, class_weight=None, ignore_index=255):
    """-> (lse [N,H,W], acc float64[4] = (weighted nll sum, #correct, #valid, #labels outside [0,C) that are not ignore_index))"""
    _dense(logits), _dense(label_u8, U8)
    n, c, h, w = logits.shape
    H, W = label_u8.shape[-2:]
    assert label_u8.numel() == n * H * W
    if pix_weight is not None:
        assert _dense(pix_weight).numel() == n * H * W
    lse = torch.empty(n, H, W, device=logits.device)
    acc = torch.zeros(4, dtype=F64, device=logits.device)
    call('pfst_ce_upsample_fwd', logits.data_ptr(), n, c, h, w, label_u8.data_ptr(), _p(pix_weight), _p(class_weight), H, W,
         ignore_index, lse.data_ptr(), acc.data_ptr(), _stream())
    return lse, acc


def ce_upsample_bwd(logits, label_u8, lse, scale, pix_weight=None, class_weight=None, ignore_index=255, out=None, accumulate=False):
    n, c, h, w = logits.shape
    H, W = label_u8.shape[-2:]
    if out is None:
        assert not accumulate
        out = torch.empty_like(logits)
    call('pfst_ce_upsample_bwd', logits.data_ptr(), n, c, h, w, label_u8.data_ptr(), _p(pix_weight), _p(class_weight), H, W,
         ignore_index, _dense(lse).data_ptr(), float(scale), _dense(out).data_ptr(), int(accumulate), _stream())
    return out


def ce_finalize(acc, numel, loss_weight):
    """-> float32[3] = (loss, acc_seg %, #invalid labels)"""
    out = torch.empty(3, device=acc.device)
    call('pfst_ce_finalize', acc.data_ptr(), float(numel), float(loss_weight), out.data_ptr(), _stream())
    return out


def pseudo_label(logits, size, threshold, want_i64=True, want_conf=False, want_prob=False):
    """-> (label int64 [N,H,W] | None, label uint8 [N,H,W], count uint64-as-int64 [1][, conf float [N,H,W]][, max prob float [N,H,W]])"""
    _dense(logits)
    n, c, h, w = logits.shape
    H, W = size
    l64 = torch.empty(n, H, W, dtype=I64, device=logits.device) if want_i64 else None
    l8 = torch.empty(n, H, W, dtype=U8, device=logits.device)
    cnt = torch.empty(1, dtype=I64, device=logits.device)
    conf = torch.empty(n, H, W, device=logits.device) if want_conf else None
    prob = torch.empty(n, H, W, device=logits.device) if want_prob else None
    call('pfst_pseudo_label', logits.data_ptr(), n, c, h, w, H, W, float(threshold), _p(l64), l8.data_ptr(), cnt.data_ptr(), _p(conf),
         _p(prob), _stream())
    res = (l64, l8, cnt)
    if want_conf:
        res += (conf,)
    if want_prob:
        res += (prob,)
    return res


def label_presence(label_u8):
    _dense(label_u8, U8)
    pres = torch.empty(256, dtype=torch.int32, device=label_u8.device)
    call('pfst_label_presence', label_u8.data_ptr(), label_u8.numel(), pres.data_ptr(), _stream())
    return pres


def class_mask(gt_u8, classes):
    """classes: int32 [N, K] device tensor (entries < 0 = padding) -> mask uint8 [N,1,H,W]"""
    _dense(gt_u8, U8), _dense(classes, torch.int32)
    n = gt_u8.shape[0]
    hw = gt_u8.numel() // n
    mask = torch.empty_like(gt_u8)
    call('pfst_class_mask', gt_u8.data_ptr(), classes.data_ptr(), classes.shape[1], mask.data_ptr(), n, hw, _stream())
    return mask


def class_mix(img, trg_img, gt_u8, pseudo_u8, mask_u8, conf_count, want_i64=False, trg_weight=None):
    _dense(img), _dense(trg_img), _dense(gt_u8, U8), _dense(pseudo_u8, U8), _dense(mask_u8, U8), _dense(conf_count, I64)
    n, c, h, w = img.shape
    assert trg_img.shape == img.shape and gt_u8.numel() == n * h * w == pseudo_u8.numel() == mask_u8.numel()
    mimg = torch.empty_like(img)
    mlbl = torch.empty(n, 1, h, w, dtype=U8, device=img.device)
    mlbl64 = torch.empty(n, 1, h, w, dtype=I64, device=img.device) if want_i64 else None
    mw = torch.empty(n, h, w, device=img.device)
    call('pfst_class_mix', img.data_ptr(), trg_img.data_ptr(), gt_u8.data_ptr(), pseudo_u8.data_ptr(), mask_u8.data_ptr(),
         conf_count.data_ptr(), _p(trg_weight), mimg.data_ptr(), mlbl.data_ptr(), _p(mlbl64), mw.data_ptr(), n, c, h * w, _stream())
    return mimg, mlbl, mlbl64, mw


# ---------------------------------------------------------------- PFGSTLoss pieces
SIM_TYPES = {'cosine': 0, 'gaussian': 1}
SRC_LOSS_TYPES = {'mean_std': 0, 'margin': 1, 'margin2': 2}


def sim_map(feat, dil, sim_type='cosine', sigma=30.0):
    _dense(feat)
    n, c, h, w = feat.shape
    sim = torch.empty(n, 9, h, w, device=feat.device)
    norm = torch.empty(n, h, w, device=feat.device)
    call('pfst_sim_map', feat.data_ptr(), n, c, h, w, dil, SIM_TYPES[sim_type], float(sigma), sim.data_ptr(), norm.data_ptr(), _stream())
    return sim, norm


def sim_map_bwd(feat, sim, norm, gsim, dil, out=None, accumulate=False, sim_type='cosine', sigma=30.0):
    n, c, h, w = feat.shape
    if out is None:
        assert not accumulate
        out = torch.empty_like(feat)
    coef = torch.empty(n * 10 * h * w, dtype=F32, device=feat.device) if sim_type == 'cosine' else None
    call('pfst_sim_map_bwd', _dense(feat).data_ptr(), _dense(sim).data_ptr(), _dense(norm).data_ptr(), _dense(gsim).data_ptr(),
         n, c, h, w, dil, SIM_TYPES[sim_type], float(sigma), _dense(out).data_ptr(), int(accumulate), _p(coef), _stream())
    return out


def src_sim_losses(sim, gt_u8, dil, w_pos, w_neg, w_pos_std=0.0, w_neg_std=0.0, loss_type='mean_std', margin=(0.5, 0.5), src_perc=None):
    """-> (losses float32[4] (mean_std) or [2] used of 4 (margin / margin2), gsim [N,9,H,W]).
    src_perc: only the int(n * src_perc) smallest positive / largest negative similarities count (pfgst_loss.py:98-102)"""
    n, _, h, w = sim.shape
    hg, wg = gt_u8.shape[-2:]
    lt = SRC_LOSS_TYPES[loss_type]
    sel = None
    if src_perc is not None:
        sel = torch.empty(lib().pfst_src_sim_select_bytes() // 8 + 1, dtype=I64, device=sim.device)       # 8-byte aligned scratch
        call('pfst_src_sim_select', _dense(sim).data_ptr(), _dense(gt_u8, U8).data_ptr(), n, h, w, hg, wg, dil, float(src_perc),
             sel.data_ptr(), _stream())
    stats = torch.empty(6, dtype=F64, device=sim.device)
    call('pfst_src_sim_stats', _dense(sim).data_ptr(), _dense(gt_u8, U8).data_ptr(), n, h, w, hg, wg, dil, lt, float(margin[0]),
         float(margin[1]), stats.data_ptr(), _p(sel), _stream())
    gsim = torch.empty_like(sim)
    losses = torch.empty(4, device=sim.device)
    call('pfst_src_sim_grad', sim.data_ptr(), gt_u8.data_ptr(), n, h, w, hg, wg, dil, lt, float(margin[0]), float(margin[1]),
         stats.data_ptr(), float(w_pos), float(w_neg), float(w_pos_std), float(w_neg_std), gsim.data_ptr(), losses.data_ptr(), _p(sel),
         _stream())
    return losses, gsim


def softmax_down(logits, ds):
    _dense(logits)
    n, c, h, w = logits.shape
    H, W = int(h // ds), int(w // ds)
    prob = torch.empty(n, c, H, W, device=logits.device)
    call('pfst_softmax_down', logits.data_ptr(), n, c, h, w, ds, prob.data_ptr(), H, W, _stream())
    return prob


def trg_valid_mask(gt_u8, mix_mask_u8, hw, dil):
    n = gt_u8.shape[0]
    hg, wg = gt_u8.shape[-2:]
    valid = torch.empty(n, 1, hw[0], hw[1], dtype=U8, device=gt_u8.device)
    all9 = torch.empty(n, 1, hw[0], hw[1], dtype=U8, device=gt_u8.device)
    cnt = torch.empty(1, dtype=I64, device=gt_u8.device)
    call('pfst_trg_valid_mask', _dense(gt_u8, U8).data_ptr(), _dense(mix_mask_u8, U8).data_ptr(), n, hw[0], hw[1], hg, wg, dil,
         valid.data_ptr(), all9.data_ptr(), cnt.data_ptr(), _stream())
    return valid, all9, cnt


def sim_topk_loss(ema_sim, prob, valid, count, dil, top_k, w_pos, w_neg, want_sim_grad=False):
    """top_k None / 0 = all nine pairs.  -> (losses float32[2], gP [N,9,H,W][, d losses / d ema_sim [N,9,H,W]])"""
    top_k = int(top_k or 0)
    n, c, h, w = prob.shape
    gP = torch.empty(n, 9, h, w, device=prob.device)
    gS = torch.empty(n, 9, h, w, device=prob.device) if want_sim_grad else None
    acc = torch.empty(2, dtype=F64, device=prob.device)
    call('pfst_sim_topk_loss', _dense(ema_sim).data_ptr(), _dense(prob).data_ptr(), _dense(valid, U8).data_ptr(), count.data_ptr(),
         n, c, h, w, dil, top_k, float(w_pos), float(w_neg), gP.data_ptr(), acc.data_ptr(), _p(gS), _stream())
    out = torch.empty(2, device=prob.device)
    call('pfst_sim_loss_finalize', acc.data_ptr(), count.data_ptr(), top_k, float(w_pos), float(w_neg), out.data_ptr(), _stream())
    return (out, gP, gS) if want_sim_grad else (out, gP)


def cross_prob_bwd_(dlogits, prob, gP, dil, ds, unfold_grad=False):
    n, c, H, W = prob.shape
    h, w = dlogits.shape[-2:]
    call('pfst_cross_prob_bwd', _dense(prob).data_ptr(), _dense(gP).data_ptr(), n, c, H, W, dil, ds, int(unfold_grad),
         _dense(dlogits).data_ptr(), h, w, _stream())
    return dlogits


# ---------------------------------------------------------------- flat-arena optimiser steps
def ema_update_(teacher_flat, student_flat, alpha):
    assert teacher_flat.numel() == student_flat.numel()
    call('pfst_ema_update', _dense(teacher_flat).data_ptr(), _dense(student_flat).data_ptr(), teacher_flat.numel(), float(alpha), _stream())


def adamw_step_(p, g, m, v, lr, betas, eps, weight_decay, step, grad_scale=1.0):
    n = p.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n
    call('pfst_adamw_step', _dense(p).data_ptr(), _dense(g).data_ptr(), _dense(m).data_ptr(), _dense(v).data_ptr(), n, float(lr),
         float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step), float(grad_scale), _stream())


# ---------------------------------------------------------------- strong augmentation
def color_jitter_(img, params, mean3, std3, denorm=True):
    _dense(img)
    n, c, h, w = img.shape
    assert c == 3 and tuple(params.shape) == (n, 8)
    call('pfst_color_jitter', img.data_ptr(), _dense(params).data_ptr(), _dense(mean3).data_ptr(), _dense(std3).data_ptr(),
         n, h * w, int(denorm), _stream())
    return img


def gaussian_blur(img, taps_y, taps_x, reach):
    _dense(img)
    n, c, h, w = img.shape
    assert taps_y.shape[0] == n and taps_x.shape[0] == n
    tmp, out = torch.empty_like(img), torch.empty_like(img)
    call('pfst_gaussian_blur', img.data_ptr(), tmp.data_ptr(), out.data_ptr(), _dense(taps_y).data_ptr(), taps_y.shape[1],
         _dense(taps_x).data_ptr(), taps_x.shape[1], n, c, h, w, int(reach), _stream())
    return out


# ---------------------------------------------------------------- evaluation
def confusion_hist_(hist, pred_u8, label_u8, num_classes, ignore_index=255):
    """hist: int64 [3*C] device tensor, accumulated in place."""
    _dense(pred_u8, U8), _dense(label_u8, U8), _dense(hist, I64)
    assert pred_u8.numel() == label_u8.numel() and hist.numel() == 3 * num_classes
    call('pfst_confusion_hist', pred_u8.data_ptr(), label_u8.data_ptr(), pred_u8.numel(), num_classes, ignore_index,
         hist.data_ptr(), _stream())
    return hist
