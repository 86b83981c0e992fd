"""Minimal explicit autograd for the PFST train step: a reverse-ordered tape of closures over HIP ops.

Why not torch.autograd: the graph is static, every backward op is one of our own kernels, gradients
of branching activations are accumulated by the kernels themselves (`accumulate` epilogues), channel
concatenations are never materialised (producers write into slices, consumers' data-gradients are
read through slices), and both student graphs (source + mixed pass) run backward in one sweep like
the reference's single `total_loss.backward()` (rsiseg/models/uda/pfgst.py:344)."""
import torch

from . import hip_ops as ops


class Var:
    """An activation (dense NCHW tensor, or a channel slice of a parent Var) with a gradient slot.

    `bn`: set on the output of a conv -> BN layer (layers.BnBackwardCtx).  The data-gradient launch that COMPLETES this Var's
    gradient can then emit the layer's BatchNorm-backward sums from its epilogue.  Which launch completes it is static: the tape
    runs closures in reverse recording order, so the LAST writer is the FIRST consumer recorded in forward -- `claim_first_use()`.
    The claim is enforced at run time: after a writer has declared itself final (`grad_target(final=True)`), any further
    `grad_target()` on the Var raises instead of silently invalidating the fused sums."""
    __slots__ = ('data', '_grad', 'requires_grad', 'parent', 'c0', 'c1', 'bn', '_claimed', '_sealed', 'amax', 'lazy', 'gate_consumer', 'pending', 'coef_table',
                 'lazy_norelu', 'dual')

    def __init__(self, data, requires_grad=False, parent=None, c0=0, c1=0):
        self.data = data
        self._grad = None
        self.requires_grad = requires_grad
        self.parent, self.c0, self.c1 = parent, c0, c1
        self.bn = None
        self._claimed = self._sealed = False
        # (pre, coef): a conv -> BN -> ReLU output that is never materialised -- data is None, the ONE consumer (a depthwise layer, the stem's
        # max-pool) normalises the pre-BN tensor `pre` with coef[c] = (mean, invstd, sc, sh) as it loads it (layers.conv_bn_act(defer=True))
        self.lazy = None
        self.lazy_norelu = False      # lazy, and the deferred layer has NO ReLU (a downsample conv -> BN: read only as bn_apply's residual operand)
        self.amax = None          # device slot with max |data| (scale of the two-piece fp16 split, layers.CONV_MATH == 'f16x3'), set on first use
        # A residual block's identity branch: dL/d(this Var) = [its other consumers' gradients] + g where the block's final ReLU let the
        # sum through.  `gate_consumer`: set in forward by the first consumer (= the last gradient writer) when its data-gradient launch can add
        # such a gated tensor in its epilogue (layers.conv_bn_act); `pending` = (g, mask): set in backward by the residual layer INSTEAD of
        # writing g * gate into the gradient buffer, taken over by that consumer (take_pending) -- or, if anything else asks for the gradient
        # first, written out after all (_flush_pending: same values, one more pass).
        self.gate_consumer = False
        self.pending = None
        # a downsample layer's output (conv -> BN, no ReLU; read only as the residual of the block's bn3): what bn3's backward needs to run BOTH
        # layers' BatchNorm backward in one reduction + one apply pass (layers.FUSE_BN_BWD_DUAL); 'done' = (dL/dpre, its amax group) once it has
        self.dual = None
        # a concat buffer whose writers leave their PRE-normalisation outputs in its slices (layers.conv_bn_act(defer='slice')): [C, 4] rows
        # (mean, invstd, sc, sh) per channel, filled slice by slice; the owner turns the buffer into a lazy Var for its single consumer
        self.coef_table = None

    def claim_first_use(self):
        """forward: called by every consumer that may fuse; True for the first caller only (= the last gradient writer)"""
        first = not self._claimed
        self._claimed = True
        return first and self.parent is None

    def grad_unwritten(self):
        return self.parent is None and self._grad is None

    def take_pending(self):
        p, self.pending = self.pending, None
        return p

    def _flush_pending(self):
        if self.pending is not None:
            (g, mask), self.pending = self.pending, None
            buf, acc = self.grad_target()
            ops.relu_gate_(buf, g, mask, accumulate=acc)

    @property
    def grad(self):
        if self.parent is not None:
            pg = self.parent.grad
            return None if pg is None else pg[:, self.c0:self.c1]
        self._flush_pending()
        return self._grad

    def grad_target(self, final=False):
        """-> (buffer, accumulate): where a backward kernel must write this Var's gradient.  final: the caller completes the
        gradient (and fuses the owner's BatchNorm-backward sums): later writers are an error."""
        if self._sealed:
            raise RuntimeError('gradient written after the launch that fused its BatchNorm-backward sums (a consumer did not '
                               'claim_first_use() in forward order)')
        self._flush_pending()
        self._sealed = bool(final)
        if self.parent is not None:
            buf, acc = self.parent.grad_target_full()
            return buf[:, self.c0:self.c1], acc
        if self._grad is None:
            like = self.data if self.lazy is None else self.lazy[0]
            self._grad = torch.empty(like.shape, dtype=like.dtype, device=like.device)
            return self._grad, False
        return self._grad, True

    def grad_target_full(self):
        # slices of a concat buffer are written by different producers' backward: zero once, accumulate after
        if self._grad is None:
            self._grad = torch.empty(self.data.shape, dtype=self.data.dtype, device=self.data.device)
            ops.fill_(self._grad, 0.0)
        return self._grad, True

    def free_grad(self):
        self._grad = None

    def slice(self, c0, c1):
        return Var(self.data[:, c0:c1], self.requires_grad, parent=self, c0=c0, c1=c1)


class Tape:
    """Closures recorded in forward order, executed in reverse; None tape = no-grad (teacher) mode.

    Every closure may carry a `tag` (op name + the Vars it reads / writes); `observer(tag, phase)` is called with phase
    'pre' / 'post' around each tagged closure during backward().  The product never sets an observer; the per-link backward
    parity test uses it to feed every closure of the network AS WIRED the oracle's upstream gradient."""

    def __init__(self, observer=None):
        self.fns = []
        self.observer = observer

    def record(self, fn, tag=None):
        self.fns.append((fn, tag))

    def pop(self):
        return self.fns.pop()

    def backward(self):
        obs = self.observer
        while self.fns:
            fn, tag = self.fns.pop()
            if obs is not None and tag is not None:
                obs(tag, 'pre')
                fn()
                obs(tag, 'post')
            else:
                fn()
        # weight gradients queued on the side stream (layers.WGRAD_STREAM) are ordered before whatever the caller does next on this stream
        from . import layers
        layers.join_side_stream()


class ParamArena:
    """All parameters of one network in ONE contiguous fp32 device buffer (+ an equally laid out
    gradient buffer).  EMA, AdamW and the RCCL all-reduce each act on the flat buffer once."""

    def __init__(self, named_params, device, with_grad=True):
        self.names, self.offsets, self.shapes = [], {}, {}
        off = 0
        for name, p in named_params:
            self.names.append(name)
            self.offsets[name] = off
            self.shapes[name] = tuple(p.shape)
            off += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device) if with_grad else None
        for name, p in named_params:
            v = self.view(self.data, name)
            v.copy_(p.data)
            p.data = v
            if with_grad:
                p.grad = self.view(self.grad, name)
            p._pfst_arena = self

    def view(self, flat, name):
        o = self.offsets[name]
        shp = self.shapes[name]
        n = 1
        for s in shp:
            n *= s
        return flat[o:o + n].view(shp)

    def zero_grad(self):
        ops.fill_(self.grad, 0.0)
