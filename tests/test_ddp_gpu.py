"""Data-parallel train step rehearsed with 2 ranks sharing the one GPU of the test box (gloo backend moves the CUDA
buffers; on the 8-GPU node the same code runs over RCCL): the reduced student gradient equals the mean of the
per-rank single-process gradients, every rank ends the step with identical student weights, teachers stay local."""
import os
import random

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import assert_live_target_side, seeded_pfgst_state, to_dev, uda_cfg

pytestmark = pytest.mark.gpu


def _one_step(rank, seed_rng=True):
    import pfst_amd  # noqa: F401
    from oracle import pfst_oracle as O
    from pfst_amd.optim import build_optimizer
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import synth_batch
    model = UDA.build(uda_cfg(threshold=0.30))
    both, _, _ = seeded_pfgst_state(O, 9)
    model.load_state_dict(both, strict=False)
    model.cuda()
    opt = build_optimizer(model, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01))
    random.seed(52 + 2 * rank); np.random.seed(52 + 2 * rank)
    out = model.train_step(to_dev(synth_batch(2, 128, 6, seed=1234 + rank), 'cuda'), opt)
    assert_live_target_side(out['log_vars'])      # every rank's step runs the whole loss graph (64-px label blocks, synthetic.py)
    a = model.student_arena
    return out, a.grad.clone().cpu(), a.data.clone().cpu(), model._teacher_arena.data.clone().cpu()


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    out, grad, weights, teacher = _one_step(rank)
    torch.save(dict(log=out['log_vars'], grad=grad, weights=weights, teacher=teacher), os.path.join(outdir, f'r{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.parametrize('overlap', ['1', '0'])
def test_two_rank_data_parallel_step(tmp_path, overlap):
    """overlap=1: the bucketed reducer launched from marker closures of the source pass's backward (dist.GradReducer);
    overlap=0: one reduction after the sweep.  Same result."""
    os.environ['PFST_DDP_OVERLAP'] = overlap         # read at import in the spawned ranks
    port = 29600 + (os.getpid() % 1000) + (7 if overlap == '1' else 0)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    os.environ.pop('PFST_DDP_OVERLAP', None)
    r0 = torch.load(tmp_path / 'r0.pt', weights_only=False)
    r1 = torch.load(tmp_path / 'r1.pt', weights_only=False)
    assert torch.equal(r0['grad'], r1['grad']), 'all ranks must hold the same reduced gradient'
    assert torch.equal(r0['weights'], r1['weights']), 'all ranks must end the step with identical student weights'
    assert r0['log'] == r1['log'], 'log_vars are averaged over ranks'
    # single-process references for the two shards
    (o0, g0, _, _), (o1, g1, _, _) = _one_step(0), _one_step(1)
    mean = 0.5 * (g0 + g1)
    err = float((r0['grad'].double() - mean.double()).norm() / mean.double().norm())
    assert err < 1e-4, err
    for k in o0['log_vars']:
        assert abs(r0['log'][k] - 0.5 * (o0['log_vars'][k] + o1['log_vars'][k])) < 1e-4 * max(1.0, abs(r0['log'][k]))


def _nccl_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    x = torch.arange(1000, dtype=torch.float32, device='cuda')
    dist.all_reduce(x, op=dist.ReduceOp.AVG)            # the op pfst_amd.dist uses on RCCL
    t = torch.tensor([3.5], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    # the bucketed reducer's async AVG all-reduces on slices of one flat buffer, behind work queued on the current stream
    from pfst_amd import dist as pdist
    flat = torch.zeros(3_000_000, device='cuda')
    flat += torch.arange(3_000_000, dtype=torch.float32, device='cuda') % 97         # producer kernel still in the queue
    red = pdist.GradReducer(flat)
    red.ready(2_000_000)
    flat[:2_000_000] *= 2.0                                                           # the sweep keeps writing the rest
    red.ready(1_000_000)
    red.finish()
    torch.cuda.synchronize()
    expect = torch.arange(3_000_000, dtype=torch.float32) % 97
    expect[:2_000_000] *= 2.0
    ok_red = bool(torch.equal(flat.cpu(), expect))
    ok = bool(torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float32))) and float(t) == 3.5 and ok_red
    open(os.path.join(outdir, 'nccl_ok'), 'w').write(str(ok))
    dist.destroy_process_group()


def test_rccl_backend_ops_used_by_the_exchange(tmp_path):
    """Single-rank RCCL group: the collectives / reduce ops the N>1 path relies on exist and run on this stack."""
    port = 29700 + (os.getpid() % 1000)
    mp.spawn(_nccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert open(tmp_path / 'nccl_ok').read() == 'True'
