"""N>1 path on CPU: world_size-2 gloo run of the exchange step (flat gradient mean in slices + packed log vector)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pfst_amd import dist as pdist
    assert pdist.is_distributed()
    n = 100_003                                     # not a multiple of the slice size
    g = torch.Generator().manual_seed(7)
    base = torch.randn(n, generator=g)
    flat = base * (rank + 1)                        # rank r holds (r+1)*base -> mean = base*(1+2)/2
    pdist.allreduce_mean_(flat, slice_elems=4096)
    ok1 = torch.allclose(flat, base * 1.5, rtol=1e-6, atol=1e-7)
    packed = torch.tensor([1.0 + rank, 10.0 * (rank + 1), 0.5])
    red = pdist.reduce_log_vector(packed)
    ok2 = torch.allclose(red, torch.tensor([1.5, 15.0, 0.5]))
    pdist.check_same_keys(['a', 'b', 'c'])
    # the overlapped, bucketed reducer: tails of the arena are launched as they become final, the rest at finish()
    arena = base * (rank + 1)
    red = pdist.GradReducer(arena, min_bucket=1000)
    red.ready(90_000)            # [90000, n) goes out
    red.ready(89_500)            # smaller than a bucket: deferred
    arena[:60_000] += 1.0        # "the backward sweep" keeps writing the part that is not final yet
    red.ready(60_000)            # [60000, 90000)
    assert len(red.pending) == 2 and red.hi == 60_000
    red.finish()
    expect = base * 1.5
    expect[:60_000] += 1.0
    ok1 = ok1 and torch.allclose(arena, expect, rtol=1e-6, atol=1e-6) and not red.pending
    q.put((rank, bool(ok1), bool(ok2)))
    dist.destroy_process_group()


def test_gradient_and_log_exchange_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, True), (1, True, True)]


def _bcast_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import torch.nn as nn
    from pfst_amd import dist as pdist
    from pfst_amd.data import epoch_indices
    torch.manual_seed(100 + rank)                   # --diff_seed: every rank initialises differently ...
    m = nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8), nn.Conv2d(8, 4, 1))
    m[1].running_mean.add_(rank + 1.0)
    before = torch.cat([t.flatten().float() for t in m.state_dict().values()])
    pdist.broadcast_module_state_(m)                # ... and continues from rank 0's parameters AND buffers
    after = torch.cat([t.flatten().float() for t in m.state_dict().values()])
    gathered = [torch.empty_like(after) for _ in range(world)]
    dist.all_gather(gathered, after)
    same = all(torch.equal(g, gathered[0]) for g in gathered)
    changed = not torch.equal(before, after)
    # the sampler shards one shared permutation: the ranks' index sets are disjoint and cover the data set
    idx = torch.tensor(epoch_indices(10, world, rank, epoch=2, seed=3))
    allidx = [torch.empty_like(idx) for _ in range(world)]
    dist.all_gather(allidx, idx)
    cover = sorted(torch.cat(allidx).tolist()) == list(range(10))
    q.put((rank, bool(same), bool(changed), bool(cover)))
    dist.destroy_process_group()


def test_rank0_state_broadcast_and_sampler_sharding_world2():
    """tools/train.py: after building (and loading) the model every rank takes rank 0's state before the per-rank random streams may
    diverge -- what MMDistributedDataParallel's constructor does in the reference (rsiseg/apis/train.py:104-112); ADVICE r1."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, False, True), (1, True, True, True)]        # rank 1 changed, rank 0 did not; all equal afterwards
