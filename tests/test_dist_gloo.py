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
