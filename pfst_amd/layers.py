"""Parameter holders with the reference's module/state_dict names and the fused layer forwards
(conv -> train-mode BN -> ReLU [+ residual]) on the HIP ops, each recording its backward closure.

Naming mirrors mmcv's ConvModule / DepthwiseSeparableConvModule (`.conv`, `.bn`, `.depthwise_conv`,
`.pointwise_conv`) because checkpoints key on it (SURVEY.md §8b "Checkpoint layout")."""
import math

import torch
import torch.nn as nn

from . import hip_ops as ops
from .engine import Var

import os

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
_BN_EVAL = False
# Arithmetic of the dense convolutions' GEMMs (forward, data gradient, 1x1 / Winograd-domain weight gradients).  All three give fp32
# results to fp32 round-off; every parity test runs under each of them, and tests/test_fullsize_gpu.py compares them layer by layer at
# b=8 x 1024^2 against fp64.  gfx950's fp32-input MFMA runs at 1/16 of the bf16 / fp16 rate, hence the splits:
#   'f16x3'  -- TWO scaled fp16 pieces per operand, three fp16 MFMAs per product, fp32 accumulate (csrc/conv_f16x3.hip): the default
#               (round 3).  Every operand tensor is scaled by the power of two its absolute maximum implies (device scalars published by
#               the producing kernels).  Measured error vs fp64 at or below the fp32-input MFMA's on every layer; covers contractions
#               over whole 32-channel blocks with more than 32 output rows (33 ... 64: the 64-row tile), the other layers run as under 'bf16x6';
#   'bf16x6' -- three bf16 pieces per operand, the six piece-products >= 2^-16 (csrc/conv_split.hip); PFST_CONV_MATH=bf16x6;
#   'f32'    -- fp32-input MFMA (v_mfma_f32_32x32x2_f32); PFST_CONV_MATH=f32.  Layers whose channel count is not a multiple of 16
#               (the 3-/10-band stems, the classifiers' data gradient) use it in every mode.
CONV_MATH = os.environ.get('PFST_CONV_MATH', 'f16x3')


def _split_mode():
    return CONV_MATH in ('bf16x6', 'f16x3')


def amax_of(v):
    """device slot group with max |v.data| of a Var: published by the kernel that produced the tensor (bn_apply, the Winograd
    transforms) or, failing that, computed once here; kept on the Var -- an activation usually feeds several GEMMs"""
    if v.amax is None:
        if v.lazy is not None:
            raise RuntimeError('a deferred conv -> BN -> ReLU output (conv_bn_act(defer=True): never materialised) has a second consumer; '
                               'only the single depthwise layer / max-pool that normalises on load may read it')
        v.amax = ops.absmax(v.data)
    return v.amax


def _amax_target(yv, dev):
    """where the kernel producing Var `yv` publishes max |y| in f16x3 mode: the Var's own slot group; for a channel slice of a concat
    buffer the buffer's group if its owner set one up, else none (whoever reads such a buffer through an f16x3 GEMM then computes the
    maximum itself, amax_of)."""
    if CONV_MATH != 'f16x3':
        return None
    if yv.parent is not None:
        # a concat buffer's group exists only where its owner makes EVERY writer of the buffer publish into it (the ASPP head's 2560-channel
        # concat: four normalisation passes + the image-pool broadcast): a slice writer then adds its maximum to the shared group
        return yv.parent.amax
    if yv.amax is None:
        yv.amax = ops.amax_slots(dev)
    return yv.amax
# bf16x6 mode: the 1x1 and Winograd-domain weight gradients run on the K-quad split kernel (1.5x the fp32-MFMA one); the rare direct
# 3x3 / strided ones stay on fp32 MFMA (the generic split kernel is slower than fp32 MFMA)
# Module-level switches below are NOT environment switches (round 5: every settled A/B lost its PFST_* variable): the tests flip the ones whose
# 'off' wiring they compare against (per-link backward test, fold on / off tests); the product runs the values written here.
FUSE_BN_STATS = True         # batch statistics out of the producing kernel's epilogue (False: the stand-alone pfst_bn_stats pass)
# Winograd F(m x m,3x3) for the wide stride-1 3x3 layers (csrc/conv_winograd.hip): fp32 results, 4x (m = 4, default) or 2.25x
# (PFST_WINO_TILE=2) fewer MACs.
# Threshold on Cin*Cout from tools/wino_microbench.py / bench.py: Winograd wins for fprop, dgrad and (with the transformed input
# kept from the forward pass and the grouped launch) the weight gradient from 128x128 channels up with F(4x4) (x1.4 at layer2,
# x3.0 at the head bottleneck), from 256x256 up with F(2x2); 64x64 (layer1) stays direct (x0.9).
WINOGRAD = True
_WINO_DEFAULT_CC = 128 * 128 if ops.WINO_TILE == 4 else 256 * 256
WINO_MIN_CC = _WINO_DEFAULT_CC
WINO_MIN_CC_WGRAD = _WINO_DEFAULT_CC
# BatchNorm backward: the two per-channel sums come out of the epilogue of the data-gradient launch that completes dL/dy
# (csrc/conv_epilogue.h, pfst_bnb_fuse_t) wherever that launch is a K-quad implicit GEMM; elsewhere the two-pass kernels run.
FUSE_BN_BWD = True
# residual blocks: the identity branch's gradient, dL/d(block output) gated by the block's final ReLU, is added in the epilogue of the launch that
# completes dL/d(block input) (conv1's f16x3 data gradient; a downsample layer's BatchNorm backward takes it as its gated dy) instead of
# being written by the bn3 layer's BatchNorm backward and read back -- one write of the block's widest tensor less per block.
# False: pfst_bn_backward writes it (dres)
FUSE_RES_GATE = True
# f16x3: the weight gradients the whole-line kernel does not take (direct stride-1 3x3, 1x1 with <= 64 output channels) on the K-quad kernel
# with both operands split as they are staged (csrc/conv_wgrad_q.hip)
# ... but only where that launch is MFMA-bound: with a short contraction (K = Cout * taps of the consuming conv) the data gradient is
# itself HBM-bound (layer1: 2 * 64 flop per 8 bytes written + accumulated), and reading the pre-BN tensor there costs what the
# reduction pass would have cost (measured: fusing everywhere moves 11 ms/step out of pfst_bn_backward and 10 ms into the GEMMs)
FUSE_BN_BWD_MIN_K = 512
# the bf16x6 data-gradient kernel carries the same epilogue; measured (b=8, 4-step runs): 439.2 ms with every launch fused, 437.7 ms with none --
# the cost is not VALU-vs-MFMA contention but the longer workgroup lifetime, in either arithmetic
# (round 5, with the mask-gated epilogue: 256 instead of 512 moves 2.1 ms out of bn_backward and 1.7 ms into the GEMMs: -0.4 ms per step, left at 512)
FUSE_BN_BWD_MIN_K_SPLIT = 512
# the same sums out of the Winograd output transform of a data-gradient launch (bn1 behind a Winograd conv2, the aux head's / ASPP bottleneck's
# inputs where they have a single producer ...): saves the reduction pass of those layers -- -0.95 ms per step in the same-box A/B taken while the
# switch read the environment (profiles/r05_ab_bnb_wino.txt)
FUSE_BN_BWD_WINO = True
# a residual layer's gate inside the fused sums: from bn_apply's 1-bit mask of y instead of y (a compile-time variant of the f16x3 epilogue;
# -0.8 ms per step, profiles/r05_ab_bnb_mask.txt)
BNB_GATE_FROM_MASK = True


class BnBackwardCtx:
    """what the launch completing a conv -> BN layer's output gradient needs to emit that layer's BatchNorm-backward sums"""
    __slots__ = ('pre', 'y', 'coef', 'relu', 'partials', 'slots', 'mask')

    def __init__(self, pre, y, coef, relu, mask=None):
        self.pre, self.y, self.coef, self.relu = pre, y, coef, relu
        self.mask = mask                 # bn_apply's ReLU bitmask of y (residual layers): the f16x3 epilogue gates with its bits instead of reading y
        self.partials, self.slots = None, 0


class bn_eval:
    """`with bn_eval():` -- BatchNorm uses running statistics (model.eval() at test time, apis/test.py:60-116)."""

    def __enter__(self):
        global _BN_EVAL
        self.prev, _BN_EVAL = _BN_EVAL, True

    def __exit__(self, *a):
        global _BN_EVAL
        _BN_EVAL = self.prev


class Conv2dP(nn.Module):
    """Weights of an nn.Conv2d (+ K-major packings for the implicit GEMM)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, groups=1, bias=False):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        assert groups in (1, cin)
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        fan_out = cout * k * k // groups
        nn.init.normal_(self.weight, 0.0, math.sqrt(2.0 / fan_out))     # kaiming_normal_(fan_out, relu)
        self.wf = self.wd = None        # packed copies, refreshed by repack()
        self.w6f = self.w6d = None      # bf16x3-split packed copies (CONV_MATH == 'bf16x6')
        self.uf = self.ud = None        # Winograd transform-domain filters

    split_f = split_d = False
    f16_f = f16_d = wino_f16 = False          # this layer's fprop / dgrad / Winograd-domain GEMMs run on the f16x3 kernel
    w4f = w4d = w_amax = uf_amax = ud_amax = None
    pf = pd = None                            # plain transform-domain filter sets (WeightBatch keeps them per layer)
    wino = False
    saved_v = None

    def _wino_eligible(self):
        return (WINOGRAD and self.k == 3 and self.stride == 1 and self.groups == 1
                and self.padding == self.dilation and self.cin % 16 == 0 and self.cout % 16 == 0
                and self.cin * self.cout >= WINO_MIN_CC)

    @property
    def depthwise(self):
        return self.groups > 1

    def wino_wgrad_ok(self, h, w):
        return self.wino and self.cin * self.cout >= WINO_MIN_CC_WGRAD and ops.wino_tiles(h, w, self.dilation) % 4 == 0

    def fprop(self, xd, out=None, bias=None, want_stats=False, keep=False, x_amax=None, bnl=None, want_minmax=False):
        """keep: training forward -- the Winograd path keeps its transformed input for the weight gradient (self.saved_v);
        x_amax: the slot with max |xd| when the caller has it (f16x3 layers; computed here otherwise);
        bnl: xd is the PRE-normalisation output of the layer feeding this one, coef [C, 4] its record: normalised on load (Winograd path only);
        want_minmax (f16x3 GEMM, with want_stats): the statistics also carry the output's per-channel (min, max) partials"""
        self.saved_v = None
        if self.wino and bias is None:
            keep_v = keep and self.wino_wgrad_ok(xd.shape[2], xd.shape[3])
            assert bnl is None or keep_v or not keep, 'a normalise-on-load input must leave its transform behind for the weight gradient'
            res = ops.wino_conv(xd, self.uf, self.cout, self.dilation, out=out, keep_v=keep_v, want_stats=want_stats,
                                u_amax=self.uf_amax if self.wino_f16 else None, x_amax=x_amax if self.wino_f16 else None, bnl=bnl,
                                want_minmax=want_minmax)
            if keep_v:
                self.saved_v = res[-1]
                res = res[:-1] if len(res) > 2 else res[0]
            return res                                     # (y, stats, slots) with want_stats, else y
        assert bnl is None or (self.fprop_bnl_ok() and bias is None and x_amax is not None), \
            'only the Winograd input transform, the 256-row 1x1 f16x3 GEMM (and the depthwise / max-pool kernels) normalise on load'
        if self.f16_f:
            return ops.conv_fprop_f16x3(xd, self.w4f, self.w_amax, x_amax if x_amax is not None else ops.absmax(xd), self.cout, self.k,
                                        self.stride, self.dilation, self.padding, bias=bias, out=out, want_stats=want_stats,
                                        want_minmax=want_minmax, bnl=bnl)
        if self.split_f:
            return ops.conv_fprop_split(xd, self.w6f, self.cout, self.k, self.stride, self.dilation, self.padding, bias=bias, out=out,
                                        want_stats=want_stats)
        return ops.conv_fprop(xd, self.wf, self.cout, self.k, self.stride, self.dilation, self.padding, bias=bias, out=out,
                              want_stats=want_stats)

    def fprop_bnl_ok(self):
        """the forward GEMM can normalise its input as it loads it (ops.conv_fprop_f16x3(bnl=...)) and the weight gradient likewise
        (ops.conv_wgrad_f16x3_(bnl=...)): a 1x1 f16x3 layer on the 256-row tile -- every Bottleneck conv3"""
        return (CONV_MATH == 'f16x3' and self.f16_f and not self.wino and not self.depthwise and self.bias is None and self.cout > 64
                and ops.conv_fprop_bnl_ok(self.cin, self.cout, self.k, self.stride, self.padding))

    def can_fuse_bn_backward(self):
        """the data gradient runs on the K-quad implicit-GEMM kernel with whole row tiles (its epilogue can emit the sums)"""
        if self.wino:                     # the Winograd output transform carries the sums (ops.wino_conv(bnb=...)): one more read of the pre-BN tensor in
            return FUSE_BN_BWD and FUSE_BN_BWD_WINO            # an HBM-bound kernel instead of the reduction pass's two
        min_k = FUSE_BN_BWD_MIN_K_SPLIT if (self.split_d or self.f16_d) else FUSE_BN_BWD_MIN_K
        if self.f16_d and self.cin % 128 != 0:
            return False                  # the f16x3 kernel's fused epilogue needs whole 128-row tiles
        return (FUSE_BN_BWD and not self.depthwise and self.cout % 16 == 0
                and self.cin % ops.bnb_tile_rows(self.cin) == 0 and self.cout * self.k * self.k >= min_k)

    def wgrad_f16q_ok(self, h, w):
        """the weight gradient runs on the f16x3 K-quad kernel (ops.conv_wgrad_f16q_): stride-1 'same' 3x3 outside the Winograd dispatch (the
        stems, layer1 conv2).  The kernel also takes 1x1 layers; those with <= 64 output channels are HBM-bound and measured no faster on it
        (0.225 vs 0.205 ms for layer1 conv1 on the bf16x6 kernel): they stay where they are"""
        if CONV_MATH != 'f16x3' or self.depthwise or self.stride != 1 or self.cout < 16:
            return False
        if self.wino_wgrad_ok(h, w):
            return False
        return self.k == 3 and w % 16 == 0 and self.dilation <= 8 and self.padding == self.dilation

    def dgrad_can_gate(self, in_hw):
        """the data-gradient launch can add a ReLU-gated tensor in its epilogue (ops.conv_dgrad_f16x3(gate=...))"""
        return FUSE_RES_GATE and self.f16_d and not self.wino and not self.depthwise and ops.dgrad_gate_ok(self.cin, in_hw)

    def dgrad(self, dy, in_hw, out, accumulate, bn=None, dy_amax=None, gate=None):
        """bn: BnBackwardCtx of the layer that produced this conv's input, when this launch completes that gradient;
        dy_amax: slot group with max |dy| (f16x3 layers; computed here when the producer did not publish it);
        gate: (g, mask) added where the mask has the bit (Var.pending of a residual block's input; only where dgrad_can_gate)"""
        assert gate is None or (self.f16_d and not self.wino and not accumulate)
        if self.wino:
            if bn is not None:           # the output transform emits the BatchNorm-backward sums of the layer that owns `out` (no residual there)
                assert bn.y is None
                _, bn.partials, bn.slots = ops.wino_conv(dy, self.ud, self.cin, self.dilation, out=out, accumulate=accumulate,
                                                         u_amax=self.ud_amax if self.wino_f16 else None,
                                                         x_amax=dy_amax if self.wino_f16 else None, bnb=(bn.pre, bn.coef, bn.relu))
                return out
            return ops.wino_conv(dy, self.ud, self.cin, self.dilation, out=out, accumulate=accumulate,
                                 u_amax=self.ud_amax if self.wino_f16 else None, x_amax=dy_amax if self.wino_f16 else None)
        if self.f16_d:
            amax = dy_amax if dy_amax is not None else ops.absmax(dy)
            if bn is not None:
                _, bn.partials, bn.slots = ops.conv_dgrad_f16x3(dy, self.w4d, self.w_amax, amax, self.cin, in_hw, self.k, self.stride,
                                                                self.dilation, self.padding, out=out, accumulate=accumulate,
                                                                bnb=(bn.pre, bn.y, bn.coef, bn.relu, bn.mask), gate=gate)
                return out
            return ops.conv_dgrad_f16x3(dy, self.w4d, self.w_amax, amax, self.cin, in_hw, self.k, self.stride, self.dilation,
                                        self.padding, out=out, accumulate=accumulate, gate=gate)
        if self.split_d:
            if bn is not None:
                _, bn.partials, bn.slots = ops.conv_dgrad_split(dy, self.w6d, self.cin, in_hw, self.k, self.stride, self.dilation,
                                                                self.padding, out=out, accumulate=accumulate,
                                                                bnb=(bn.pre, bn.y, bn.coef, bn.relu))
                return out
            return ops.conv_dgrad_split(dy, self.w6d, self.cin, in_hw, self.k, self.stride, self.dilation, self.padding,
                                        out=out, accumulate=accumulate)
        if bn is not None:
            _, bn.partials, bn.slots = ops.conv_dgrad(dy, self.wd, self.cin, in_hw, self.k, self.stride, self.dilation, self.padding,
                                                      out=out, accumulate=accumulate, bnb=(bn.pre, bn.y, bn.coef, bn.relu))
            return out
        return ops.conv_dgrad(dy, self.wd, self.cin, in_hw, self.k, self.stride, self.dilation, self.padding,
                              out=out, accumulate=accumulate)

    def repack(self, need_dgrad, batch=None, rec=None):
        """batch: a WeightBatch collecting the f16x3 images of the whole network for its three launches (the other packings, a few
        small layers, are launched here either way); rec: a list that receives those other launches as (function, args) -- the model replays
        them on later steps instead of walking its layers again (EncoderDecoder.repack_weights)"""
        if self.depthwise:
            return

        def launch(fn, *a):
            if rec is not None:
                rec.append((fn, a))
            return fn(*a)
        self.wino = self._wino_eligible()
        f16 = CONV_MATH == 'f16x3'
        self.wino_f16 = self.f16_f = self.f16_d = False
        if self.wino:
            split = _split_mode()                   # transform-domain GEMMs on a split kernel: split-packed filter sets
            # (the transform-domain GEMMs have no 64-row tile: more than 64 rows in both directions)
            self.wino_f16 = (f16 and ops.f16x3_eligible(self.cin, self.cout) and self.cout > 64
                             and (not need_dgrad or (ops.f16x3_eligible(self.cout, self.cin) and self.cin > 64)))
            n = (ops.WINO_TILE + 2) ** 2 * (self.weight.numel() // 9) * ((4 if self.wino_f16 else 6) if split else 1)
            dt = torch.uint8 if split else torch.float32
            if self.uf is None or self.uf.device != self.weight.device or self.uf.dtype != dt or self.uf.numel() != n:
                self.uf = torch.empty(n, dtype=dt, device=self.weight.device)
                self.ud = None
            if need_dgrad and self.ud is None:
                self.ud = torch.empty(n, dtype=dt, device=self.weight.device)
            if self.wino_f16 and batch is not None:
                batch.add_wino(self, need_dgrad)
            elif self.wino_f16:
                _, _, self.uf_amax, self.ud_amax = ops.wino_pack_weight_f16(self.weight.data, True, need_dgrad, self.uf,
                                                                            self.ud if need_dgrad else None)
            else:
                pack = ops.wino_pack_weight_split if split else ops.wino_pack_weight
                launch(pack, self.weight.data, True, need_dgrad, self.uf, self.ud if need_dgrad else None)
            if self.bias is None:
                return                    # the direct-convolution packings are not needed
        self.f16_f = f16 and ops.f16x3_eligible(self.cin, self.cout, self.k)
        self.f16_d = f16 and need_dgrad and ops.f16x3_eligible(self.cout, self.cin, self.k)
        # the fp32 K-major images only for the directions that run on the fp32-input MFMA kernels (every layer under 'f32'; the stems
        # and the classifiers' data gradient otherwise): 106 small launches per step less in the split modes
        fp32_f = not (self.f16_f or (_split_mode() and self.cin % 16 == 0))
        fp32_d = need_dgrad and not (self.f16_d or (_split_mode() and self.cout % 16 == 0))
        if fp32_f or fp32_d:
            if self.wf is None or self.wf.device != self.weight.device:
                self.wf = torch.empty(self.k * self.k * self.cin, self.cout, device=self.weight.device)
                self.wd = None
            if fp32_d and self.wd is None:
                self.wd = torch.empty(self.k * self.k * self.cout, self.cin, device=self.weight.device)
            launch(ops.pack_weight, self.weight.data, fp32_f, fp32_d, self.wf if fp32_f else None, self.wd if fp32_d else None)
        if self.f16_f or self.f16_d:
            nbytes = 4 * self.weight.numel()
            if self.f16_f and (self.w4f is None or self.w4f.device != self.weight.device):
                self.w4f = torch.empty(nbytes, dtype=torch.uint8, device=self.weight.device)
            if self.f16_d and (self.w4d is None or self.w4d.device != self.weight.device):
                self.w4d = torch.empty(nbytes, dtype=torch.uint8, device=self.weight.device)
            if batch is not None:
                batch.add_direct(self)
            else:
                _, _, self.w_amax = ops.pack_weight_f16x2(self.weight.data, self.f16_f, self.f16_d, self.w4f if self.f16_f else None,
                                                          self.w4d if self.f16_d else None)
        self.split_f = _split_mode() and self.cin % 16 == 0 and not self.f16_f
        self.split_d = _split_mode() and self.cout % 16 == 0 and need_dgrad and not self.f16_d
        if self.split_f or self.split_d:
            nbytes = 6 * self.weight.numel()
            if self.split_f and (self.w6f is None or self.w6f.device != self.weight.device):
                self.w6f = torch.empty(nbytes, dtype=torch.uint8, device=self.weight.device)
            if self.split_d and (self.w6d is None or self.w6d.device != self.weight.device):
                self.w6d = torch.empty(nbytes, dtype=torch.uint8, device=self.weight.device)
            launch(ops.pack_weight_split, self.weight.data, self.split_f, self.split_d, self.w6f if self.split_f else None,
                   self.w6d if self.split_d else None)


def repack_key(need_dgrad):
    """everything Conv2dP.repack's decisions depend on besides the layer itself: a model replays its recorded packing launches while this (and
    its weight buffers) stay the same"""
    return (CONV_MATH, bool(need_dgrad), WINOGRAD, WINO_MIN_CC, WINO_MIN_CC_WGRAD, ops.WINO_TILE, ops.F16X3_MIN_ROWS, WeightBatch.enabled)


class WeightBatch:
    """The f16x3 weight images of a whole network in three launches per step -- zero the slot groups, one preparation launch (absolute
    maxima of the directly convolved layers; filter transform + per-set maxima of the Winograd layers), one packing launch -- instead
    of two to five 5-12 us launches per convolution (~290 per step for student + teacher, 1 % of the b=8 step).  The job tables
    (ops.WeightJobTable) are rebuilt only when a buffer or a mode changed.  enabled = False: the per-layer launches (tests)."""
    enabled = True

    def __init__(self):
        self.key = None
        self.items = []

    def begin(self):
        self.items = []

    def add_direct(self, conv):
        self.items.append((conv, 0, False))

    def add_wino(self, conv, need_dgrad):
        n = (ops.WINO_TILE + 2) ** 2 * conv.cout * conv.cin
        dev = conv.weight.device
        if conv.pf is None or conv.pf.device != dev or conv.pf.numel() != n:      # plain transform-domain sets, resident per layer
            conv.pf, conv.pd = torch.empty(n, device=dev), None
        if need_dgrad and conv.pd is None:
            conv.pd = torch.empty(n, device=dev)
        self.items.append((conv, ops.WINO_TILE, need_dgrad))

    def flush(self):
        if not self.items:
            return
        key = tuple((id(c), m, dg, c.weight.data_ptr(), c.f16_f, c.f16_d) + tuple(ops._p(getattr(c, a)) for a in ('w4f', 'w4d', 'uf', 'ud', 'pf', 'pd'))
                    for c, m, dg in self.items)
        if key != self.key:
            self._build()
            self.key = key
        for c, attr, view in self.views:      # a per-layer repack() in between may have replaced them
            setattr(c, attr, view)
        self.slots.zero_()
        self.prep.run('pfst_weight_prep_batched')
        self.pack.run('pfst_conv_pack_weight_f16x2_batched')

    def replay(self):
        """the launches of the last flush() again (same layers, same buffers: the caller checked)"""
        if not self.items:
            return
        for c, attr, view in self.views:
            setattr(c, attr, view)
        self.slots.zero_()
        self.prep.run('pfst_weight_prep_batched')
        self.pack.run('pfst_conv_pack_weight_f16x2_batched')

    def _build(self):
        dev = self.items[0][0].weight.device
        groups = sum(1 if m == 0 else (m + 2) ** 2 * (2 if dg else 1) for _, m, dg in self.items)
        self.slots = torch.empty(groups * ops.AMAX_SUB, device=dev)
        self.views, prep, pack, at = [], [], [], 0

        def take(g):
            nonlocal at
            v = self.slots[at * ops.AMAX_SUB:(at + g) * ops.AMAX_SUB]
            at += g
            return v
        for c, m, dg in self.items:
            dims = dict(Cout=c.cout, Cin=c.cin)
            if m == 0:
                a = take(1)
                self.views.append((c, 'w_amax', a))
                prep.append(dict(src=c.weight.data, amax_f=a, T=c.k * c.k, **dims))
                pack.append(dict(src=c.weight.data, dst_f=c.w4f if c.f16_f else None, dst_d=c.w4d if c.f16_d else None, amax_f=a,
                                 T=c.k * c.k, **dims))
            else:
                x = (m + 2) ** 2
                af, ad = take(x), take(x) if dg else None
                self.views += [(c, 'uf_amax', af), (c, 'ud_amax', ad)]
                prep.append(dict(src=c.weight.data, dst_f=c.pf, dst_d=c.pd if dg else None, amax_f=af, amax_d=ad, T=9, m=m, **dims))
                pack.append(dict(src=c.pf, dst_f=c.uf, amax_f=af, T=1, sets=x, **dims))
                if dg:
                    pack.append(dict(src=c.pd, dst_d=c.ud, amax_f=ad, T=1, sets=x, **dims))
        self.prep, self.pack = ops.WeightJobTable(prep, 0, dev), ops.WeightJobTable(pack, 1, dev)


class BatchNorm2dP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self._pending_batches:
            self.num_batches_tracked += self._pending_batches
            self._pending_batches = 0
        super()._save_to_state_dict(destination, prefix, keep_vars)


class ConvModule(nn.Module):
    """conv -> BN(train) -> ReLU, `bias = not with_norm` (mmcv ConvModule semantics)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, groups=1):
        super().__init__()
        self.conv = Conv2dP(cin, cout, k, stride, padding, dilation, groups, bias=False)
        self.bn = BatchNorm2dP(cout)

    def forward(self, x, tape, relu=True, residual=None, out=None, post_scale=None, defer=False):
        return conv_bn_act(x, self.conv, self.bn, tape, relu, residual, out, post_scale, defer)


class DepthwiseSeparableConvModule(nn.Module):
    def __init__(self, cin, cout, k, padding=0, dilation=1):
        super().__init__()
        self.depthwise_conv = ConvModule(cin, cin, k, 1, padding, dilation, groups=cin)
        self.pointwise_conv = ConvModule(cin, cout, 1)

    def forward(self, x, tape, out=None, post_scale=None, defer=False):
        # FOLD_BN_DWSEP: the depthwise stage's conv -> BN -> ReLU output is never written -- the pointwise layer's f16x3 GEMM and weight gradient
        # normalise the depthwise output as they load it (the depthwise kernel emits the (min, max) partials of the predicted maximum)
        h, w = (x.data if x.lazy is None else x.lazy[0]).shape[-2:]
        fold = FOLD_BN_DWSEP and self.pointwise_conv.conv.fprop_bnl_ok() and (h * w) % 128 == 0
        return self.pointwise_conv(self.depthwise_conv(x, tape, defer='amax' if fold else False), tape, out=out, post_scale=post_scale, defer=defer)


# the atrous depthwise branches of the ASPP head as ONE launch each way (csrc/dwconv.hip, pfst_dwconv3x3_multi_*); False: per branch
FUSE_ASPP_DW = True
# a conv -> BN -> ReLU output whose ONLY consumer can normalise on load (a depthwise layer: sep_bottleneck[0] -> [1]; the stem's max-pool) is
# never written: the consumer reads the pre-BN tensor.  False: every normalised tensor is materialised (per-link test)
DEFER_BN_APPLY = True
# Bottleneck conv1 -> bn1 -> ReLU -> conv2 with conv2 on the Winograd path: the input transform normalises as it loads, y1 is never written
# (round 5).  Under f16x3 the transform writes V pre-split and needs max |y1| BEFORE y1 exists: conv1's epilogue emits per-channel
# (min, max) partials and bn_finalize_partials predicts the maximum exactly.  False: bn_apply writes y1 (per-link test).  Same-box A/B of the
# switch while it still read the environment: profiles/r05_ab_fold_bn_wino.txt
FOLD_BN_WINO = True
# the ASPP head's concat (image pool | 1x1 branch | three atrous pointwise layers) -> bottleneck 3x3 through the Winograd domain: the four
# conv -> BN -> ReLU writers leave their PRE-normalisation outputs in the concat's slices and their (mean, invstd, sc, sh) rows in the
# buffer's coefficient table; the bottleneck's input transform normalises per channel as it loads (identity rows for the image-pool slice).
# False: every slice is written normalised (per-link test)
FOLD_BN_CONCAT = True
# Bottleneck conv2 -> bn2 -> ReLU -> conv3 (1x1): conv3's f16x3 GEMM normalises conv2's PRE-normalisation output between load and split
# (coefficient rows in LDS), its weight gradient the same rows (coefficients in registers), BatchNorm backward of bn2 reads the pre-BN tensor:
# y2 is never written.  The predicted max |y2| comes from the (min, max) partials of conv2's producer -- the Winograd output transform or, in
# layer1, the direct GEMM's epilogue.  False: bn_apply writes y2 (per-link test).  Same-box A/B while the switch read the environment:
# profiles/r05_ab_fold_bn_gemm.txt (-1.05 ms per step: bn_apply -2.2, the normalising GEMMs and weight gradients +1.2)
FOLD_BN_GEMM = True
# DepthwiseSeparableConvModule (the ASPP head's three atrous branches, sep_bottleneck[0] / [1]): depthwise conv -> BN -> ReLU -> pointwise 1x1.
# The same fold: the pointwise GEMM (up to 2048 coefficient rows in LDS) and its weight gradient normalise the depthwise kernel's output on
# load, the depthwise kernels (strip, whole-plane, three-branch) emit the (min, max) partials; the depthwise layer's BatchNorm backward never
# read y anyway (its second pass runs inside the depthwise backward).  Five tensors of 1.07-1.17 GB per pass are not written and not re-read:
# -3.0 ms per step in the same-box A/B taken while the switch read the environment (profiles/r05_ab_fold_bn_dwsep.txt: bn_apply -5.1 ms, the
# normalising GEMMs and weight gradients +2.1)
FOLD_BN_DWSEP = True
# the downsample branch of a residual block (conv 1x1 -> BN, no ReLU; resnet.py:298-303): its normalised output has one reader, the residual
# operand of the block's bn3 normalisation pass -- which applies fma(r, sc, sh) itself as it loads the branch's pre-BN tensor
# (pfst_bn_apply residual_coef).  Four tensors of 0.27-1.07 GB per pass are not written and not re-read: -2.1 ms per step in the same-box A/B
# taken while the switch read the environment (profiles/r05_ab_fold_bn_residual.txt).  False: bn_apply writes them
FOLD_BN_RESIDUAL = True
# BatchNorm backward of a stage's first block: bn3 and the downsample branch's BN receive the same gated gradient -- one reduction over (g, pre3,
# pre_d) and one apply pass writing both input gradients (pfst_bn_backward_dual) on the four largest tensors
# of a pass: 8 N instead of 10 N per block).  False: layer by layer.  -1.25 ms of kernel time, -0.4 ms on the step in the same-box A/B taken while
# the switch read the environment (profiles/r05_ab_bn_bwd_dual.txt)
FUSE_BN_BWD_DUAL = True
# depthwise conv -> BN -> ReLU layers: the second pass of BatchNorm backward is applied by the depthwise backward kernel while it stages its
# operands (dL/dpre is never written)


def dwsep_branches(x, mods, tape, outs, pool=None, defer=False):
    """DepthwiseSeparableConvModules `mods` applied to the SAME input x (the ASPP head's atrous branches, sep_aspp_head.py:63-77), results
    into the concat slices `outs`.  Where the fused kernels cover the shape the depthwise stages run as one launch -- every plane of x is
    staged once for all branches -- and their backward as one launch after the branches' BatchNorm-backward passes: x read once, every
    branch's gradient once, dL/dx written once (per branch: 3 x (dy + x + old dx + dx)).  Each branch keeps its own BatchNorm (statistics
    from the launch's per-branch partials), its pointwise layer is the ordinary ConvModule.  Same results as the per-branch path: the
    forward bit for bit, the input gradient in a different summation order (branch sum in registers instead of through memory).
    pool: a dict the image-pool branch shares with this call -- on the fused path pool['mean'] receives the plane means of x (the
    AdaptiveAvgPool2d(1) of that branch, from the same pass over x) and the fused backward adds pool['grad'] (set by that branch's backward,
    which runs earlier) / (H W) to dL/dx; left untouched on the per-branch path (the caller then pools and broadcasts itself)."""
    convs = [m.depthwise_conv.conv for m in mods]
    dils = [c.dilation for c in convs]
    fused = (FUSE_ASPP_DW and len(mods) > 1 and x.parent is None and all(c.depthwise and c.k == 3 and c.stride == 1 and c.padding == c.dilation
                                                                          for c in convs) and ops.dwconv_multi_ok(x.data, dils))
    if not fused:
        return [m(x, tape, out=o, defer=defer) for m, o in zip(mods, outs)]
    xd = x.data
    want_stats = FUSE_BN_STATS and not _BN_EVAL
    # FOLD_BN_DWSEP: the branches' normalised depthwise outputs are never written, the pointwise GEMMs normalise on load (see the flag)
    n_, c_, h_, w_ = xd.shape
    fold = (FOLD_BN_DWSEP and DEFER_BN_APPLY and CONV_MATH == 'f16x3' and want_stats and (h_ * w_) % 128 == 0
            and all(m.pointwise_conv.conv.fprop_bnl_ok() for m in mods))
    if pool is not None:
        res, pool['mean'] = ops.dwconv_multi(xd, [c.weight.data for c in convs], dils, want_stats=want_stats, want_mean=True, want_minmax=fold)
    else:
        res = ops.dwconv_multi(xd, [c.weight.data for c in convs], dils, want_stats=want_stats, want_minmax=fold)
    dpres, bnbs = [None] * len(mods), [None] * len(mods)
    if tape is not None:
        x.claim_first_use()

        def bwd_dw():                                # recorded FIRST: runs after the branches' BatchNorm-backward closures below
            buf, acc = x.grad_target()
            ops.dwconv_multi_bwd_([c.weight.grad for c in convs], xd, dpres, [c.weight.data for c in convs], dils, buf, accumulate=acc,
                                  mean_grad=None if pool is None else pool.pop('grad', None), bnb=bnbs if bnbs[0] is not None else None)
            for i in range(len(dpres)):
                dpres[i] = bnbs[i] = None
        tape.record(bwd_dw, dict(op='dwconv_multi', x=x, convs=convs))
    ys = []
    for i, m in enumerate(mods):
        bn = m.depthwise_conv.bn
        pre, st, slots = res[i]
        if _BN_EVAL:
            assert tape is None, 'eval-mode BN is inference only'
            mean, invstd, coef = bn.running_mean, torch.rsqrt(bn.running_var + BN_EPS), None
        else:
            want_coef = (tape is not None and FUSE_BN_BWD) or fold
            gb = dict(gamma=bn.weight.data, beta=bn.bias.data) if want_coef else {}
            n, c, h, w = pre.shape
            pred_amax = ops.amax_slots(pre.device) if fold else None
            if want_stats:
                out3 = ops.bn_finalize_partials(st, slots, c, n * h * w, bn.running_mean, bn.running_var, BN_MOMENTUM, BN_EPS,
                                                predict_amax=pred_amax, relu=True, **gb)
            else:
                out3 = ops.bn_stats(pre, bn.running_mean, bn.running_var, BN_MOMENTUM, BN_EPS, **gb)
            mean, invstd = out3[:2]
            coef = out3[2] if want_coef else None
            bn._pending_batches += 1
        yv = Var(None, tape is not None)
        if fold:
            yv.lazy, yv.amax = (pre, coef, bn), pred_amax          # never written: the pointwise GEMM and its weight gradient normalise on load
        else:
            yv.data = ops.bn_apply(pre, mean, invstd, bn.weight.data, bn.bias.data, True, None, amax=_amax_target(yv, pre.device))
        if tape is not None:
            if coef is not None and FUSE_BN_BWD:
                yv.bn = BnBackwardCtx(pre, None, coef, True)       # the pointwise layer's data gradient may emit this layer's sums

            def bwd_bn(i=i, bn=bn, yv=yv, pre=pre, mean=mean, invstd=invstd):
                part, nslots = (yv.bn.partials, yv.bn.slots) if yv.bn is not None else (None, 0)
                # only the sums here; the fused backward of the branches applies the second pass while it stages (dy, pre)
                rec = ops.bn_backward_sums(yv.grad, pre, mean, invstd, bn.weight.data, bn.bias.data, bn.weight.grad, bn.bias.grad,
                                           partials=part, slots=nslots)
                dpres[i], bnbs[i] = yv.grad, (pre, rec)
                yv.free_grad()
            tape.record(bwd_bn, dict(op='dw_bn_act', bn=bn, conv=convs[i], x=x, out=yv))
        ys.append(m.pointwise_conv(yv, tape, out=outs[i], defer=defer))
    return ys


def conv_forward(x, conv, tape, out=None):
    """bare convolution (used by conv_seg); returns Var"""
    if x.lazy is not None:
        raise RuntimeError('conv_forward on a deferred conv -> BN -> ReLU output (conv_bn_act(defer=True)): only a depthwise conv_bn_act or the '
                           'max-pool normalises on load')
    xd = x.data
    if conv.depthwise:
        assert conv.k == 3 and conv.stride == 1 and conv.padding == conv.dilation
        y = ops.dwconv(xd, conv.weight.data, conv.dilation, out=out)
    else:
        y = conv.fprop(xd, out=out, bias=None if conv.bias is None else conv.bias.data, keep=tape is not None,
                       x_amax=amax_of(x) if (conv.f16_f or conv.wino_f16) else None)
    saved_v = None if conv.depthwise else conv.saved_v
    yv = Var(y, tape is not None)
    if tape is not None:
        final = x.claim_first_use()

        def bwd():
            dy = yv.grad
            conv_backward(x, conv, dy, saved_v, final)
            yv.free_grad()
        tape.record(bwd, dict(op='conv', conv=conv, x=x, out=yv))
    return yv


# Stream-level overlap (default since round 5): weight gradients on a high-priority side stream beside the
# BatchNorm-backward / data-gradient chain (with the wgrad kernels capped to 2 workgroups per CU so the chain's HBM-bound kernels find
# room), and the teacher's forward pass forked beside the student's source pass.  Same-process rotating A/B on one box
# (tools/ab_streams.py, profiles/r05_ab_streams.txt): 291.0 ms single stream, 288.6 forked teacher, 288.7 side-stream weight
# gradients, 286.3 both.  Per-kernel durations measured with events then include time-sharing, so bench.py measures its roofline leg
# in a single-stream region of the same run (set_overlap(False, False)).  PFST_WGRAD_STREAM=0 / PFST_FORK_TEACHER=0: one stream.
WGRAD_STREAM = os.environ.get('PFST_WGRAD_STREAM', '1') == '1'
FORK_TEACHER = os.environ.get('PFST_FORK_TEACHER', '1') == '1'
# the step boundary (uda.PFGST): the optimizer step queued before the host blocks on the step's log values, the student's weight images
# packed behind the forked teacher pass -- the device has work while the host crosses from one step into the next (tools/gap_analysis.py)
STEP_BOUNDARY_OVERLAP = True
# the step's packed log values (all forward results) are copied to the host in front of the backward sweep: the blocking read at the end of
# PFGST.forward_train waits for the forward passes only and the next step's host-side start runs under the queued backward sweep
# -- where the reference reads them too (`_parse_losses` before `total_loss.backward()`, pfgst.py:338-344).  -2.2 ms per step in the same-box A/B
# taken while the switch read the environment (profiles/r05_ab_early_log_read.txt).  False: the copy is queued behind the backward sweep
EARLY_LOG_READ = True
WGRAD_STREAM_LDS_PAD = 24000
_side_stream = None
_teacher_stream = None


def set_overlap(wgrad_stream, fork_teacher):
    """switch the two overlap features at run time (bench.py's `alt_streams` leg, tests)"""
    global WGRAD_STREAM, FORK_TEACHER
    WGRAD_STREAM, FORK_TEACHER = bool(wgrad_stream), bool(fork_teacher)
    ops.set_wgrad_lds_pad(WGRAD_STREAM_LDS_PAD if WGRAD_STREAM else 0)


def teacher_stream():
    global _teacher_stream
    if _teacher_stream is None:
        _teacher_stream = torch.cuda.Stream()
    return _teacher_stream


def _on_side_stream(fn, *tensors):
    """run fn (weight-gradient launches) on the current stream's side stream, ordered after everything queued on the current stream so far"""
    global _side_stream
    main = torch.cuda.current_stream()
    if _side_stream is None:
        _side_stream = {}
        ops.set_wgrad_lds_pad(WGRAD_STREAM_LDS_PAD)
    side = _side_stream.get(main.cuda_stream)
    if side is None:
        side = _side_stream[main.cuda_stream] = torch.cuda.Stream(priority=-1)      # high priority: the chain's kernels fill in around the wgrads
    side.wait_stream(main)
    with torch.cuda.stream(side):
        fn()
    for t in tensors:
        if t is not None:
            t.record_stream(side)


def join_side_stream():
    """the current stream waits for the weight gradients queued on its side stream (before the optimizer / all-reduce)"""
    if _side_stream is not None:
        main = torch.cuda.current_stream()
        side = _side_stream.get(main.cuda_stream)
        if side is not None:
            main.wait_stream(side)


def _wgrad(conv, xd, dy, saved_v, x_amax=None, dy_amax=None, x_bnl=None):
    """weight gradient of a dense convolution into conv.weight.grad (fp32 atomics): Winograd-domain, 1x1 / 3x3 K-quad or generic
    kernel; in the split modes the 1x1 and Winograd-domain products use the fp32-faithful split on the bf16 / fp16 matrix cores.
    saved_v: (transformed input, its amax slot group) kept from the forward pass; x_amax / dy_amax: slot groups when the caller has
    them (f16x3)"""
    split = _split_mode()
    f16 = split and CONV_MATH == 'f16x3' and conv.cout > 64
    v, v_amax = saved_v if saved_v is not None else (None, None)
    if conv.wino_wgrad_ok(xd.shape[2], xd.shape[3]):
        # the f16x3 Winograd-domain weight gradient pairs with the f16x3 forward pass: only a `wino_f16` layer keeps a PACKED V with
        # the slot group of its scale bound.  A Winograd layer outside the f16x3 GEMM's shapes (3x3 64 -> 256, 48 -> 512: Cin not a
        # multiple of 32, or a data gradient the kernel does not cover) kept a plain fp32 V and takes the bf16x6 product
        f16w = f16 and conv.wino_f16
        ops.wino_wgrad_(conv.weight.grad, xd, dy, conv.dilation, v=v, split=2 if f16w else split, v_amax=v_amax if f16w else None,
                        x_amax=x_amax if f16w else None, dy_amax=dy_amax if f16w else None)
    elif f16 and conv.k == 1 and conv.stride == 1 and (xd.shape[2] * xd.shape[3]) % 4 == 0:
        # x_bnl: xd is the pre-normalisation tensor the forward GEMM normalised as it loaded it (conv_bn_act(defer='amax') -> conv3)
        assert x_bnl is None or x_amax is not None
        ops.conv_wgrad_f16x3_(conv.weight.grad, xd, dy, x_amax if x_amax is not None else ops.absmax(xd),
                              dy_amax if dy_amax is not None else ops.absmax(dy), bnl=x_bnl)
        return
    elif x_bnl is not None:
        raise RuntimeError('a normalise-on-load input reached a weight-gradient kernel that reads the tensor as stored')
    elif conv.wgrad_f16q_ok(xd.shape[2], xd.shape[3]) and ops.wgrad_q_operands_ok(xd, dy):
        ops.conv_wgrad_f16q_(conv.weight.grad, xd, dy, x_amax if x_amax is not None else ops.absmax(xd),
                             dy_amax if dy_amax is not None else ops.absmax(dy), conv.k, conv.dilation)
    elif split and conv.k == 1 and conv.stride == 1 and (xd.shape[2] * xd.shape[3]) % 4 == 0:
        ops.conv_wgrad_split_(conv.weight.grad, xd, dy, conv.k, conv.stride, conv.dilation, conv.padding)
    else:
        ops.conv_wgrad_(conv.weight.grad, xd, dy, conv.k, conv.stride, conv.dilation, conv.padding)


def _dgrad_into(x, conv, dy, final, dy_amax=None):
    """data gradient of `conv` into x's gradient buffer; when this launch completes the gradient of a conv -> BN layer's output
    (final) and runs on the K-quad kernel, it also emits that layer's BatchNorm-backward sums (x.bn.partials)"""
    fuse = final and x.bn is not None and x.parent is None and conv.can_fuse_bn_backward() and (not conv.wino or x.bn.y is None)
    gate = None
    if x.pending is not None and x.grad_unwritten() and conv.dgrad_can_gate(x.data.shape[-2:]):       # (a pending gate implies a materialised x)
        gate = x.take_pending()           # the identity branch's gated gradient rides in this launch's epilogue (else grad_target writes it out)
    buf, acc = x.grad_target(final=fuse)
    conv.dgrad(dy, buf.shape[-2:], buf, acc, bn=x.bn if fuse else None, dy_amax=dy_amax, gate=gate)


def conv_backward(x, conv, dy, saved_v=None, final=False, dy_amax=None, dw_bnb=None):
    """accumulate weight/bias grads and propagate the data gradient into x; saved_v: Winograd-transformed x from forward;
    final: this conv was x's first consumer in forward = the last writer of x's gradient (Var.claim_first_use);
    dy_amax: the slot group the kernel that produced dy published max |dy| to (f16x3)"""
    xd = x.data if x.lazy is None else x.lazy[0]
    x_bnl = None if x.lazy is None else x.lazy[1]          # the depthwise layer's input is normalised on load (conv_bn_act(defer=True))
    f16w = CONV_MATH == 'f16x3' and not conv.depthwise and conv.cout > 64 and conv.k == 1 and conv.stride == 1
    wino16 = CONV_MATH == 'f16x3' and not conv.depthwise and conv.wino_f16
    f16q = not conv.depthwise and x.lazy is None and conv.wgrad_f16q_ok(xd.shape[2], xd.shape[3])
    x_amax = amax_of(x) if (f16w or f16q or (wino16 and saved_v is None)) else None
    if (f16w or f16q or wino16) and dy_amax is None:
        dy_amax = ops.absmax(dy)
    if WGRAD_STREAM and not conv.depthwise:
        wg_bnl = x_bnl if not (conv.depthwise or conv.wino) else None       # (the Winograd layers' weight gradient reads the kept transform)

        def wg():
            _wgrad(conv, xd, dy, saved_v, x_amax, dy_amax, wg_bnl)
        # (every tensor the side stream reads is recorded on it: the host drops its references when this closure returns, long before the
        # queued launch runs -- the coefficient rows of a normalise-on-load input included)
        _on_side_stream(wg, dy, xd, None if saved_v is None else saved_v[0], wg_bnl, x_amax, dy_amax,
                        None if saved_v is None else saved_v[1])          # (the slot groups are views of an arena block that is replaced every ~10 steps)
        if conv.bias is not None:
            ops.bias_grad_(conv.bias.grad, dy)
        if x.requires_grad:
            _dgrad_into(x, conv, dy, final, dy_amax)
        return
    if conv.depthwise:
        if x.requires_grad:
            buf, acc = x.grad_target()         # both gradients from one staging of dy and one read of x (3 N of traffic instead of 4 N)
            ops.dwconv_bwd_(conv.weight.grad, xd, dy, conv.weight.data, conv.dilation, buf, accumulate=acc, bnl=x_bnl, bnb=dw_bnb)
        else:
            assert x_bnl is None and dw_bnb is None          # (a depthwise layer on the network input: weight gradient only)
            ops.dwconv_wgrad_(conv.weight.grad, xd, dy, conv.dilation)
    else:
        _wgrad(conv, xd, dy, saved_v, x_amax, dy_amax, x_bnl if not conv.wino else None)
        if conv.bias is not None:
            ops.bias_grad_(conv.bias.grad, dy)
        if x.requires_grad:
            _dgrad_into(x, conv, dy, final, dy_amax)


_identity_rows = {}


def identity_coef_row(dev):
    """(mean, invstd, sc, sh) = (0, 1, 1, 0) on the device: the coefficient row under which a normalise-on-load consumer leaves a channel's
    (non-negative) values alone.  Cached per device: building it from a Python list is a pageable host-to-device copy, i.e. a host-side
    wait for the stream in the middle of every forward pass"""
    row = _identity_rows.get(dev)
    if row is None:
        row = _identity_rows[dev] = torch.tensor([0.0, 1.0, 1.0, 0.0], device=dev)
    return row


def conv_bn_act(x, conv, bn, tape, relu=True, residual=None, out=None, post_scale=None, defer=False):
    """y = [relu](BN_train(conv(x)) [+ residual]); `out` may be a channel slice of a concat buffer
    (then the returned Var is expected to be obtained from the concat Var's .slice()).
    post_scale: [N, C] factors applied to y after the ReLU -- the Dropout2d mask of the layer feeding conv_seg, folded into the
    normalisation pass and, in backward, into the BatchNorm-backward passes (one tensor round trip less each way)"""
    assert post_scale is None or residual is None
    # defer: the caller guarantees that the ONLY consumer of the result normalises on load (a depthwise layer or the stem's max-pool): the
    # normalisation pass is skipped, the returned Var carries (pre, coef) in .lazy and no data.  x.lazy: this layer IS such a consumer.
    # defer='amax': the consumer is an f16x3 GEMM operand producer (the Winograd input transform, which writes V pre-split) and needs
    # max |y| of the tensor that is never written: under f16x3 the producing GEMM emits (min, max) partials and the finalize kernel predicts it
    xd = x.data if x.lazy is None else x.lazy[0]
    x_bnl = None if x.lazy is None else x.lazy[1]
    assert x.lazy is None or not x.lazy_norelu, 'a deferred downsample output (no ReLU) is only read as a residual'
    assert x_bnl is None or conv.depthwise or (conv.wino and conv.bias is None) or conv.fprop_bnl_ok(), \
        'only a depthwise layer, the Winograd input transform, a 256-row 1x1 f16x3 GEMM (or the max-pool) reads a deferred normalisation'
    # defer='slice': `out` is a slice of a concat Var with a coefficient table (Var.coef_table): the PRE-normalisation output goes into the slice,
    # this layer's rows into the table, the predicted maximum into the concat's shared slot group
    # defer='residual' (a downsample conv -> BN layer, no ReLU): the ONLY consumer is the residual operand of the block's bn3 normalisation
    # pass, which applies fma(r, sc, sh) as it loads r (ops.bn_apply(residual_coef=...)); the returned Var is marked lazy_norelu
    as_residual = defer == 'residual'
    into_slice = slice_requested = defer == 'slice'
    need_pred = defer in ('amax', 'slice') and CONV_MATH == 'f16x3'
    if into_slice:
        defer = bool(isinstance(out, Var) and out.parent is not None and out.parent.coef_table is not None and DEFER_BN_APPLY and not _BN_EVAL and relu
                     and residual is None and post_scale is None and not conv.depthwise and not conv.wino and (not need_pred or out.parent.amax is not None))
        into_slice = defer
    elif as_residual:
        defer = bool(FOLD_BN_RESIDUAL and DEFER_BN_APPLY and not _BN_EVAL and not relu and residual is None and out is None and post_scale is None
                     and not conv.depthwise)
        as_residual = defer
    else:
        defer = bool(defer and DEFER_BN_APPLY and not _BN_EVAL and relu and residual is None and out is None and post_scale is None)
    # Batch statistics over a handful of values per channel (the ASPP image-pool branch: N x C x 1 x 1, i.e. b values) are a
    # cancellation: var = E[x^2] - mean^2 from the epilogue's fp32 partial sums of squares loses what torch's two-pass variance keeps
    # (per-link test: 1.3e-3 on that layer's backward against 6e-5 for torch-fp32).  Those tiny layers take the stand-alone statistics
    # kernel, whose sums of exact fp64 squares are exact for fp32 inputs.
    tiny = xd.shape[0] * xd.shape[2] * xd.shape[3] <= 64
    fused_stats = FUSE_BN_STATS and not _BN_EVAL and not tiny
    # producers of (min, max) partials: the f16x3 GEMM epilogue and (not into a concat slice) the Winograd output transform
    mm_producer = conv.bias is None and ((conv.f16_f and not conv.wino) or (conv.wino and not into_slice) or (conv.depthwise and not into_slice))
    if need_pred and not (defer and fused_stats and mm_producer):
        defer = need_pred = into_slice = False           # no producer of (min, max) partials here: the normalised tensor is written as usual
    if into_slice and not fused_stats:
        defer = into_slice = False
    if slice_requested and not into_slice and isinstance(out, Var) and out.parent is not None and out.parent.coef_table is not None:
        # this writer normalises its slice itself after all: identity rows, max(fma(y, 1, 0), 0) = y for its ReLU'd (non-negative) values
        assert relu
        out.parent.coef_table[out.c0:out.c1] = identity_coef_row(out.data.device)
    need_pred = need_pred and defer
    pre_out = out.data if into_slice else None           # the convolution writes straight into the concat slice
    if conv.depthwise:
        if fused_stats:                            # batch statistics come out of the producing kernel in every case
            pre, st, slots = ops.dwconv(xd, conv.weight.data, conv.dilation, want_stats=True, bnl=x_bnl, want_minmax=need_pred)
        else:
            pre = ops.dwconv(xd, conv.weight.data, conv.dilation, bnl=x_bnl)
    elif fused_stats:                              # GEMM epilogue, or the Winograd output transform
        pre, st, slots = conv.fprop(xd, out=pre_out, want_stats=True, keep=tape is not None,
                                    x_amax=amax_of(x) if (conv.f16_f or conv.wino_f16) else None, bnl=x_bnl, want_minmax=need_pred)
    else:
        pre = conv.fprop(xd, keep=tape is not None, x_amax=amax_of(x) if (conv.f16_f or conv.wino_f16) else None, bnl=x_bnl)
    saved_v = None if conv.depthwise else conv.saved_v
    if _BN_EVAL:
        assert tape is None, 'eval-mode BN is inference only'
        mean, invstd = bn.running_mean, torch.rsqrt(bn.running_var + BN_EPS)
    else:
        if pre.shape[0] * pre.shape[2] * pre.shape[3] == 1:
            # torch.nn.functional.batch_norm's own check: the image-pool BatchNorm of the ASPP head sees N x C x 1 x 1, so a per-GPU batch of
            # one cannot train; same exception type and text as the reference raises
            raise ValueError(f'Expected more than 1 value per channel when training, got input size {torch.Size(pre.shape)}')
        want_coef = (tape is not None and FUSE_BN_BWD) or defer
        gb = dict(gamma=bn.weight.data, beta=bn.bias.data) if want_coef else {}
        pred_amax = (out.parent.amax if into_slice else ops.amax_slots(pre.device)) if need_pred else None
        if fused_stats:
            n, c, h, w = pre.shape
            res = ops.bn_finalize_partials(st, slots, c, n * h * w, bn.running_mean, bn.running_var, BN_MOMENTUM, BN_EPS, predict_amax=pred_amax,
                                           relu=relu, coef_out=out.parent.coef_table[out.c0:out.c1] if into_slice else None, **gb)
        else:
            res = ops.bn_stats(pre, bn.running_mean, bn.running_var, BN_MOMENTUM, BN_EPS, **gb)
        mean, invstd = res[:2]
        coef = res[2] if want_coef else None
        bn._pending_batches += 1
    out_var = None
    if isinstance(out, Var):                       # slice of a concat Var
        out_var, out = out, out.data
    # residual layers: the ReLU gate goes to the backward pass as a bitmask (1 bit instead of 4 bytes per element, read twice)
    want_mask = tape is not None and relu and residual is not None
    # f16x3: the normalisation pass publishes max |y| for the GEMMs that will read y (no separate pass over the tensor)
    yv = out_var if out_var is not None else Var(None, tape is not None)
    if defer:
        yv.lazy, y = (pre, coef, bn), None           # no normalisation pass: the consumer applies (sc, sh) of `coef` and the ReLU as it loads `pre`
        yv.lazy_norelu = as_residual                 # (a downsample layer: no ReLU -- only bn_apply's residual operand reads it)
        if need_pred and not into_slice:
            yv.amax = pred_amax                      # max |y| of the tensor that is never written (exact: bn_finalize_partials)
    else:
        res_lazy = residual is not None and residual.lazy is not None     # a downsample layer's pre-BN output: normalised by this pass as it loads it
        assert not res_lazy or residual.lazy_norelu, 'only a deferred downsample layer (conv -> BN, no ReLU) may arrive as a lazy residual'
        y = ops.bn_apply(pre, mean, invstd, bn.weight.data, bn.bias.data, relu,
                         None if residual is None else (residual.lazy[0] if res_lazy else residual.data), out=out, want_mask=want_mask,
                         amax=_amax_target(yv, pre.device), post=post_scale, residual_coef=residual.lazy[1] if res_lazy else None)
    gate = None
    if want_mask:
        y, gate = y
    if out_var is None:
        yv.data = y
    if tape is None:
        return yv
    final = x.claim_first_use()
    if residual is not None:
        residual.claim_first_use()
    in_hw = xd.shape[-2:]
    if final and x.lazy is None and x.requires_grad and conv.dgrad_can_gate(in_hw):
        x.gate_consumer = True            # this layer's data gradient completes dL/dx and can add a gated identity-branch gradient (Var.pending)
    def need_dpre_amax():                 # f16x3: does a kernel reading dL/dpre of this layer scale it from a published maximum?
        return CONV_MATH == 'f16x3' and not conv.depthwise and ((conv.f16_d and x.requires_grad) or (conv.cout > 64 and conv.k == 1)
                                                                or conv.wino_f16 or conv.wgrad_f16q_ok(pre.shape[2], pre.shape[3]))

    if FUSE_RES_GATE and not relu and residual is None and out_var is None and post_scale is None and (not defer or as_residual) and not conv.depthwise \
            and (pre.shape[2] * pre.shape[3]) % 256 == 0:
        yv.gate_consumer = True           # a downsample layer (conv -> BN, no ReLU): its BatchNorm backward takes (g, mask) as its gated dy
        if FUSE_BN_BWD_DUAL:              # ... or rides in the bn3 backward of the block it is the residual of (same g: one reduction, one apply pass)
            yv.dual = dict(x=pre, mean=mean, invstd=invstd, bn=bn, need_amax=need_dpre_amax, done=None)
    if coef is not None and out_var is None and post_scale is None and (not defer or (need_pred and not into_slice)):
        # the launch completing dL/dy may emit this layer's BatchNorm-backward sums; ReLU gate: from y for residual layers
        # (y > 0 <=> the bitmask), else recomputed from the pre-BN tensor as bn_apply computed it
        yv.bn = BnBackwardCtx(pre, y if (relu and residual is not None) else None, coef, relu,
                              gate if (BNB_GATE_FROM_MASK and relu and residual is not None) else None)

    def bwd():
        if yv.dual is not None and yv.dual['done'] is not None:
            # a downsample layer whose BatchNorm backward already ran inside the block's bn3 backward (pfst_bn_backward_dual)
            (dpre, dpre_amax), yv.dual = yv.dual['done'], None
            conv_backward(x, conv, dpre, saved_v, final, dy_amax=dpre_amax)
            if yv.parent is None:
                yv.free_grad()
            return
        ext_gate = None
        if yv.pending is not None and yv.grad_unwritten() and yv.gate_consumer and not relu and residual is None:
            dy, ext_gate = yv.take_pending()          # downsample layer: dL/dy = g where the block's final ReLU passed; the mask is the gate
        else:
            dy = yv.grad
        dres = dacc = None
        part, nslots = (yv.bn.partials, yv.bn.slots) if yv.bn is not None else (None, 0)
        dpre_amax = ops.amax_slots(pre.device) if need_dpre_amax() else None
        if residual is not None and residual.requires_grad:
            if (FUSE_RES_GATE and gate is not None and relu and residual.gate_consumer and residual.pending is None and residual.parent is None
                    and residual.grad_unwritten()):
                dual = residual.dual if (FUSE_BN_BWD_DUAL and post_scale is None and not conv.depthwise) else None
                if dual is not None:
                    # the residual is a downsample layer's output: its BatchNorm backward and this layer's read the same gated gradient
                    ds_amax = ops.amax_slots(pre.device) if dual['need_amax']() else None
                    ds_bn = dual['bn']
                    both = ops.bn_backward_dual(
                        dy, gate,
                        dict(x=pre, mean=mean, invstd=invstd, gamma=bn.weight.data, dgamma=bn.weight.grad, dbeta=bn.bias.grad, partials=part,
                             slots=nslots, amax=dpre_amax),
                        dict(x=dual['x'], mean=dual['mean'], invstd=dual['invstd'], gamma=ds_bn.weight.data, dgamma=ds_bn.weight.grad,
                             dbeta=ds_bn.bias.grad, amax=ds_amax))
                    if both is not None:
                        dual['done'] = (both[1], ds_amax)
                        conv_backward(x, conv, both[0], saved_v, final, dy_amax=dpre_amax)
                        if yv.parent is None:
                            yv.free_grad()
                        return
                residual.pending = (dy, gate)         # dL/dresidual = dy * gate: left to the launch that completes the residual's gradient
            else:
                dres, dacc = residual.grad_target()
        # without a residual the ReLU mask is recomputed from the pre-BN tensor (one HBM read less per pass)
        ymask = y if (relu and residual is not None and gate is None) else None
        if (conv.depthwise and relu and residual is None and gate is None
                and post_scale is None and x.requires_grad):
            # the depthwise backward forms dL/dpre itself from (dy, pre) and the record of the two sums: 3 N of traffic less
            rec = ops.bn_backward_sums(dy, pre, mean, invstd, bn.weight.data, bn.bias.data, bn.weight.grad, bn.bias.grad, partials=part,
                                       slots=nslots)
            conv_backward(x, conv, dy, saved_v, final, dw_bnb=(pre, rec))
            if yv.parent is None:
                yv.free_grad()
            return
        dpre = ops.bn_backward(dy, ymask, pre, mean, invstd, bn.weight.data, bn.weight.grad, bn.bias.grad,
                               relu or ext_gate is not None, dres, bool(dacc), beta=bn.bias.data, mask=gate if ext_gate is None else ext_gate,
                               partials=part, slots=nslots, amax=dpre_amax, post=post_scale)
        conv_backward(x, conv, dpre, saved_v, final, dy_amax=dpre_amax)
        if yv.parent is None:
            yv.free_grad()
    tape.record(bwd, dict(op='conv_bn_act', conv=conv, bn=bn, x=x, residual=residual, relu=relu, out=yv))
    return yv
