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
    pdist.STEP_STATS = []
    red.stats = pdist.step_stats_begin()      # what bench.py collects per step: bucket sizes, the wait in finish()
    red.ready(90_000)            # [90000, n) goes out
    red.ready(89_500)            # smaller than a bucket: deferred
    arena[:60_000] += 1.0        # "the backward sweep" keeps writing the part that is not final yet
    red.ready(60_000)            # [60000, 90000)
    assert len(red.pending) == 2 and red.hi == 60_000
    red.finish()
    expect = base * 1.5
    expect[:60_000] += 1.0
    ok1 = ok1 and torch.allclose(arena, expect, rtol=1e-6, atol=1e-6) and not red.pending
    summ = pdist.summarize_step_stats(pdist.STEP_STATS)
    pdist.STEP_STATS = None
    ok1 = ok1 and summ['buckets'] == 3 and abs(sum(summ['bucket_MB']) - 4e-6 * n) < 0.02 and summ['exposed_allreduce_ms']['mean'] >= 0.0
    ok1 = ok1 and pdist.step_stats_begin() is None                      # nothing is recorded unless somebody collects
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


def _bn_bcast_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import torch.nn as nn
    from pfst_amd import dist as pdist
    from pfst_amd.layers import BatchNorm2dP
    torch.manual_seed(5)                            # same parameters everywhere (training keeps them in sync) ...

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = nn.Sequential(nn.Conv2d(3, 8, 1), BatchNorm2dP(8), nn.BatchNorm2d(8))       # "student"
            self.ema_model = nn.Sequential(nn.Conv2d(3, 8, 1), BatchNorm2dP(8))                       # "teacher"
    net = Net()
    g = torch.Generator().manual_seed(40 + rank)     # ... but rank-local running statistics (plain BN on rank-local batches)
    for m in net.modules():
        if hasattr(m, 'running_mean'):
            m.running_mean.copy_(torch.randn(8, generator=g))
            m.running_var.copy_(0.5 + torch.rand(8, generator=g))
    x = torch.randn(2, 3, 4, 4, generator=torch.Generator().manual_seed(9))

    def score(n):                                    # an eval-mode forward through the running statistics
        y = n.model[0](x)
        for bn in (n.model[1], n.model[2]):
            y = (y - bn.running_mean.view(1, -1, 1, 1)) / (bn.running_var.view(1, -1, 1, 1) + 1e-5).sqrt()
        return y
    before = score(net).clone()
    weights_before = net.model[0].weight.clone()
    n_bn = pdist.broadcast_bn_buffers_(net)
    after = score(net)
    stats = torch.cat([torch.cat([m.running_mean, m.running_var]) for m in net.modules() if hasattr(m, 'running_mean')])
    gathered = [torch.empty_like(stats) for _ in range(world)]
    dist.all_gather(gathered, stats)
    outs = [torch.empty_like(after) for _ in range(world)]
    dist.all_gather(outs, after.detach())
    same_stats = all(torch.equal(t, gathered[0]) for t in gathered)
    same_out = all(torch.equal(t, outs[0]) for t in outs)              # every rank now evaluates rank 0's model
    rank0_unchanged = torch.equal(before, after) if rank == 0 else not torch.equal(before, after)
    q.put((rank, n_bn, bool(same_stats), bool(same_out), bool(rank0_unchanged), bool(torch.equal(weights_before, net.model[0].weight))))
    dist.destroy_process_group()


def test_bn_buffer_broadcast_before_distributed_eval_world2():
    """evaluation.build_eval_fn at world > 1: rank 0's running_mean / running_var reach every rank first (eval_hooks.py:95-107), so
    the ranks' image shards are scored by ONE model -- the single-rank evaluation of rank 0's checkpoint."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bn_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, 3, True, True, True, True), (1, 3, True, True, True, True)]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a torch.distributed.run environment (VERDICT r2 missing #1): the parent makes no GPU call,
    starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child (tools/dist_train.sh:8-17) and relays rank 0's
    line.  --dry-run shows the exact child command; --rendezvous-only runs the launch path for real (gloo here, RCCL on a node
    with one GPU per rank) up to one all-reduce."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '20', '--warmup', '5', '--dry-run'],
                         capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])['cmd']
    i = cmd.index(os.path.join(root, 'bench.py'))
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=8' in cmd[:i]
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    assert cmd[i + 1:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']          # the ranks get the bench arguments unchanged
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--rendezvous-only'],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout                                              # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec['rendezvous'] == 'ok' and rec['rccl_ranks'] == 2 and rec['n_gpus'] == 2
