"""Data-parallel exchange of the PFST step: ONE flat gradient arena + ONE packed log vector per step.

Reference: mmcv MMDistributedDataParallel's bucketed gradient all-reduce fired from `total_loss.backward()`
(rsiseg/apis/train.py:104-112, pfgst.py:344) and the 20 tiny all-reduces of `_parse_losses` (base.py:205-220).
Here: the student gradient already lives in one contiguous 174.3 MB fp32 buffer, so the exchange is a few large
RCCL all-reduces over xGMI (slices keep RCCL's staging bounded); the EMA teacher is rank-local and never reduced.
Backend-agnostic on purpose (nccl == RCCL on the GPUs, gloo in the CPU tests)."""
import torch
import torch.distributed as dist

import os

SLICE_ELEMS = 16 * 1024 * 1024      # 64 MB per collective
# bucketed all-reduce overlapped with the backward sweep (GradReducer); PFST_DDP_OVERLAP=0: one reduction after the sweep
OVERLAP_ALLREDUCE = os.environ.get('PFST_DDP_OVERLAP', '1') == '1'
# smallest bucket the overlapped reducer sends on its own: xGMI is point-to-point (ring all-reduce: per-link bound, ~20-50 us of latency per
# collective on 8 ranks), so a 174 MB gradient travels as ~10 collectives of >= 16 MB rather than one per layer
BUCKET_ELEMS = int(float(os.environ.get('PFST_DDP_BUCKET_MB', '16')) * (1 << 20) / 4)


# PFST_DDP_FORCE=1: run the exchange (reducer, packed log vector, key check) in an initialised process group of ONE rank too -- the only
# way to put RCCL's asynchronous collectives behind a real train step on a single-GPU box (tests/test_ddp_gpu.py)
FORCE_EXCHANGE = os.environ.get('PFST_DDP_FORCE', '0') == '1'


# Step statistics of the exchange (bench.py, tests): a list that receives one dict per train step -- sizes of the buckets the reducer launched,
# the time the main stream stalled in GradReducer.finish() (= the all-reduce the backward sweep did NOT hide; HIP events on CUDA tensors, host
# clock on gloo), and the host's wait in the step's single blocking read.  None (the default): nothing is recorded, no event is created.
STEP_STATS = None


def step_stats_begin():
    """-> the dict of the step that starts now, or None when nothing collects statistics"""
    if STEP_STATS is None:
        return None
    STEP_STATS.append(dict(bucket_elems=[], exposed_allreduce=None, host_read_s=None))
    return STEP_STATS[-1]


def resolve_step_stats(stats):
    """event pairs -> milliseconds (call after a device synchronisation); returns the list"""
    for d in stats:
        ev = d.get('exposed_allreduce')
        if isinstance(ev, tuple):
            d['exposed_allreduce'] = None
            d['exposed_allreduce_ms'] = ev[0].elapsed_time(ev[1])
        elif 'exposed_allreduce_ms' not in d:
            d['exposed_allreduce_ms'] = None
    return stats


def summarize_step_stats(stats):
    """per-rank summary for the bench line: the reducer's buckets (sizes of the LAST step: the schedule is static), exposed all-reduce and host
    read time as min / mean / max over the steps"""
    def mmm(vals, scale=1.0):
        vals = [v * scale for v in vals if v is not None]
        return None if not vals else dict(min=round(min(vals), 3), mean=round(sum(vals) / len(vals), 3), max=round(max(vals), 3))
    resolve_step_stats(stats)
    last = stats[-1] if stats else {}
    return dict(steps=len(stats), buckets=len(last.get('bucket_elems', [])), bucket_MB=[round(4e-6 * n, 2) for n in last.get('bucket_elems', [])],
                exposed_allreduce_ms=mmm([d.get('exposed_allreduce_ms') for d in stats]),
                host_read_ms=mmm([d.get('host_read_s') for d in stats], 1000.0))


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_EXCHANGE)


def allreduce_mean_(flat, group=None, slice_elems=SLICE_ELEMS):
    """In-place mean over ranks of a flat tensor (DDP gradient semantics)."""
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    assert flat.dim() == 1 and flat.is_contiguous()
    use_avg = dist.get_backend(group) == 'nccl'
    op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
    n = flat.numel()
    for beg in range(0, n, slice_elems):
        dist.all_reduce(flat[beg:min(n, beg + slice_elems)], op=op, group=group)
    if not use_avg:
        flat.mul_(1.0 / world)
    return flat


def reduce_log_vector(packed, group=None):
    """Mean over ranks of the packed per-step scalars (what `_parse_losses` does key by key)."""
    world = dist.get_world_size(group)
    if world == 1:
        return packed
    dist.all_reduce(packed, group=group)
    return packed / world


def check_same_keys(names, group=None):
    """The reference asserts that every rank logs the same number of variables (base.py:205-212) to avoid
    hangs; here the check is folded into the packed vector's length via a single int all-reduce at start-up."""
    t = torch.tensor([len(names)], dtype=torch.int64)
    if dist.get_backend(group) == 'nccl':
        t = t.cuda()
    dist.all_reduce(t, group=group)
    assert int(t.item()) == len(names) * dist.get_world_size(group), 'loss log variables are different across GPUs!'


def broadcast_module_state_(module, src=0, group=None):
    """What wrapping the model in MMDistributedDataParallel does at construction (rsiseg/apis/train.py:104-112): every rank starts
    from rank `src`'s parameters and buffers.  Without it `--diff_seed` (or any rank-dependent initialisation) would average the
    gradients of DIFFERENT replicas.  Tensors are moved through the backend's device (CUDA for nccl == RCCL, CPU for gloo)."""
    if not is_distributed():
        return
    on_gpu = dist.get_backend(group) == 'nccl'
    for _, t in sorted(module.state_dict().items()):
        if not torch.is_tensor(t):
            continue
        buf = t.detach()
        if on_gpu and not buf.is_cuda:
            buf = buf.cuda()
        elif not on_gpu and buf.is_cuda:
            buf = buf.cpu()
        buf = buf.contiguous()
        dist.broadcast(buf, src=src, group=group)
        if buf.data_ptr() != t.data_ptr():
            t.detach().copy_(buf)


def broadcast_bn_buffers_(module, src=0, group=None):
    """DistEvalHook._do_evaluate (rsiseg/core/evaluation/eval_hooks.py:95-107): BatchNorm running statistics are never reduced in
    training (plain BN, rank-local batches), so before a distributed validation pass every rank takes rank `src`'s `running_var` /
    `running_mean` -- all ranks then score their image shard with the SAME model, the one `save_checkpoint` writes from rank 0.
    The reference issues two broadcasts per BatchNorm module (2 x 140 for student + teacher); here the buffers travel as one flat
    tensor.  Returns the number of BatchNorm modules synchronised (0 when not distributed)."""
    if not is_distributed():
        return 0
    bufs = []
    for _, m in module.named_modules():
        rm, rv = getattr(m, 'running_mean', None), getattr(m, 'running_var', None)
        if torch.is_tensor(rm) and torch.is_tensor(rv):
            bufs += [rv, rm]
    if not bufs:
        return 0
    on_gpu = dist.get_backend(group) == 'nccl'
    dev = torch.device('cuda', torch.cuda.current_device()) if on_gpu else torch.device('cpu')   # the backend's device
    flat = torch.cat([b.detach().reshape(-1).to(dev, torch.float32) for b in bufs])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for b in bufs:
        n = b.numel()
        b.detach().copy_(flat[off:off + n].view(b.shape))
        off += n
    return len(bufs) // 2


class GradReducer:
    """Bucketed gradient all-reduce overlapped with the backward sweep (what MMDistributedDataParallel's reducer does in the
    reference: rsiseg/apis/train.py:104-112, fired from total_loss.backward(), pfgst.py:344).

    The student gradient lives in ONE flat arena whose layout follows the module order (backbone stem .. layer4, decode head,
    auxiliary head), and the single backward sweep runs the mixed-pass graph first and the source-pass graph last, each from the
    heads down to the stem.  So during the SOURCE pass's backward the arena becomes final from its END towards its start: marker
    closures recorded in that pass's forward (`ready(offset)`: everything at or above `offset` is final) launch the all-reduce of
    the finished tail asynchronously -- RCCL runs it on its own stream behind everything queued so far -- while the sweep
    continues; `finish()` reduces what is left and makes the current stream wait for all of it.  Mean over ranks (AVG on RCCL, SUM
    and a scale on gloo)."""

    def __init__(self, flat, group=None, min_bucket=None):
        assert flat.dim() == 1 and flat.is_contiguous()
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group)
        self.avg = dist.get_backend(group) == 'nccl'
        self.hi = flat.numel()               # everything at or above `hi` is already on its way
        self.min_bucket = BUCKET_ELEMS if min_bucket is None else min_bucket
        self.pending = []
        self.stats = None                    # the step's statistics dict (step_stats_begin), set by the caller that collects them

    def _launch(self, lo, hi):
        if hi <= lo:
            return
        from . import layers
        layers.join_side_stream()            # weight gradients queued on a side stream (opt-in overlap) must have landed
        chunk = self.flat[lo:hi]
        if self.stats is not None:
            self.stats['bucket_elems'].append(hi - lo)
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        self.pending.append((dist.all_reduce(chunk, op=op, group=self.group, async_op=True), chunk))

    def ready(self, offset):
        """gradients at arena offsets >= offset are final"""
        offset = max(0, min(int(offset), self.hi))
        if self.hi - offset >= self.min_bucket:
            self._launch(offset, self.hi)
            self.hi = offset

    def finish(self):
        self._launch(0, self.hi)
        self.hi = 0
        t0 = ev0 = None
        if self.stats is not None:
            # how long this stream stalls for collectives the sweep did not hide: on RCCL work.wait() only orders the stream behind the
            # collective, so the stall is the time between two events around the waits; on gloo wait() blocks the host
            if self.flat.is_cuda:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            else:
                import time
                t0 = time.perf_counter()
        for work, chunk in self.pending:
            work.wait()
            if not self.avg:
                chunk.mul_(1.0 / self.world)
        if ev0 is not None:
            ev1.record()
            self.stats['exposed_allreduce'] = (ev0, ev1)
        elif t0 is not None:
            import time
            self.stats['exposed_allreduce_ms'] = 1000.0 * (time.perf_counter() - t0)
        self.pending = []
        return self.flat
