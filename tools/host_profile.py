#!/usr/bin/env python3
"""Where does the HOST spend a train step?  cProfile over a few steps of the bench workload (the device runs behind the host for most of a
step; what matters is the host work between the step's blocking read of its log values and the first long kernel of the next step, during
which the device idles: tools/gap_analysis.py).

  python tools/host_profile.py [--steps 4] [--top 45]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--top', type=int, default=45)
    args = ap.parse_args()
    import bench
    from pfst_amd import hip_ops
    from pfst_amd.optim import build_optimizer, poly_lr
    from pfst_amd.presets import OPTIMIZER, workload_cfg
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import fill_state_dict, synth_batch

    dev = torch.device('cuda', 0)
    cfg, w = workload_cfg(bench.WORKLOAD)
    batch = synth_batch(w['per_gpu_batch'], w['size'], w['num_classes'], w['in_channels'], seed=1234, device=dev)
    model = UDA.build(cfg)
    fill_state_dict(model.state_dict(), 0)
    model.to(dev)
    opt = build_optimizer(model, OPTIMIZER)
    it = [0]

    def run(n):
        for _ in range(n):
            for g in opt.param_groups:
                g['lr'] = poly_lr(OPTIMIZER['lr'], it[0], cfg['max_iters'])
            model.train_step(batch, opt)
            it[0] += 1

    run(3)
    torch.cuda.synchronize()
    # the launch calls in program order with host time stamps: how long after a step's blocking read the next launches are issued
    inner = hip_ops.call
    log = []

    def stamped(name, *a):
        log.append((time.perf_counter(), name))
        return inner(name, *a)
    hip_ops.call = stamped
    ev_sync = torch.cuda.Event.synchronize

    def stamped_sync(self):                # the step's blocking read (and the label-presence read) return here
        r = ev_sync(self)
        log.append((time.perf_counter(), '<Event.synchronize returned>'))
        return r
    torch.cuda.Event.synchronize = stamped_sync
    t0 = time.perf_counter()
    run(2)
    torch.cuda.synchronize()
    hip_ops.call = inner
    torch.cuda.Event.synchronize = ev_sync
    names = [n for _, n in log]
    # the second step starts at the second pfst_ema_update
    k = [i for i, n in enumerate(names) if n == 'pfst_ema_update']
    if len(k) >= 2:
        i0 = k[1]
        print('host time stamps around the step boundary (ms relative to the last launch of the previous step):')
        base = log[i0 - 1][0]
        for t, n in log[max(0, i0 - 5):i0 + 30]:
            print(f'  {1e3 * (t - base):8.3f}  {n}')
    print(f'launches per step: {len(log) / 2:.0f}; host wall per step {1e3 * (time.perf_counter() - t0) / 2:.1f} ms')
    pr = cProfile.Profile()
    pr.enable()
    run(args.steps)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(args.top)


if __name__ == '__main__':
    main()
