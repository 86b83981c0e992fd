#!/usr/bin/env python3
"""Same-process, same-box comparison of the stream-overlap options against the single-stream default (two driver runs put
bench.py's `alt_streams` leg 2.7-3.8 % ahead of `value`, the builder's run-per-leg A/B said 0.2 %).

One model, one process; the legs are run in rotating order so that no leg is systematically the one that follows a long warm phase:
  single          -- one stream, no event brackets (what a training run executes)
  single+events   -- one stream with the dominant kernel's launches bracketed by HIP events (what bench.py's timed region does for `value`)
  fork            -- teacher forward forked beside the student's source pass
  wgrad           -- weight gradients on the high-priority side stream
  fork+wgrad      -- both (the product schedule since round 5); `+old`: without the step-boundary overlap (layers.STEP_BOUNDARY_OVERLAP)
Each leg: `--warm` untimed steps after the switch, then `--steps` timed ones.  Prints one line per leg and round, then per-leg medians.

  python tools/ab_streams.py --rounds 4 --steps 8 --warm 2"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=4)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warm', type=int, default=2)
    ap.add_argument('--legs', default='single,single+events,fork,wgrad,fork+wgrad')
    args = ap.parse_args()
    import bench
    from pfst_amd import hip_ops, layers
    from pfst_amd.optim import build_optimizer, poly_lr
    from pfst_amd.presets import OPTIMIZER, workload_cfg
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import fill_state_dict, synth_batch

    dev = torch.device('cuda', 0)
    cfg, w = workload_cfg(bench.WORKLOAD)
    b, S = w['per_gpu_batch'], w['size']
    batch = synth_batch(b, S, w['num_classes'], w['in_channels'], seed=1234, device=dev)
    model = UDA.build(cfg)
    fill_state_dict(model.state_dict(), 0)
    model.to(dev)
    opt = build_optimizer(model, OPTIMIZER)
    timer = bench.KernelTimer(hip_ops.call)
    it = [0]

    def run(n):
        for _ in range(n):
            for g in opt.param_groups:
                g['lr'] = poly_lr(OPTIMIZER['lr'], it[0], cfg['max_iters'])
            model.train_step(batch, opt)
            it[0] += 1

    def leg(name):
        wg, fk = 'wgrad' in name, 'fork' in name
        layers.set_overlap(wg, fk)
        layers.STEP_BOUNDARY_OVERLAP = 'old' not in name
        events = name.endswith('+events')
        hip_ops.call = timer.inner
        run(args.warm)
        if events:
            hip_ops.call = timer.call
            timer.records, timer.only, timer.enabled, timer.per_layer = [], bench.DOMINANT_KERNEL['f16x3'], True, False
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timer.enabled = False
        hip_ops.call = timer.inner
        timer.records = []
        layers.set_overlap(False, False)
        layers.STEP_BOUNDARY_OVERLAP = True
        return 1000.0 * dt / args.steps

    names = args.legs.split(',')
    run(3)
    res = {n: [] for n in names}
    for r in range(args.rounds):
        order = names[r % len(names):] + names[:r % len(names)]
        if r % 2:
            order = order[::-1]
        for n in order:
            ms = leg(n)
            res[n].append(ms)
            print(f'round {r} {n:14s} {ms:8.2f} ms/step  {b * 1000.0 / ms:6.3f} images/s', flush=True)
    med = {n: statistics.median(v) for n, v in res.items()}
    base = med.get('single', next(iter(med.values())))
    for n in names:
        print(f'median {n:14s} {med[n]:8.2f} ms/step  ({med[n] - base:+6.2f} vs single; min {min(res[n]):.2f} max {max(res[n]):.2f})')
    print(json.dumps({'median_ms': med, 'all_ms': res}))


if __name__ == '__main__':
    main()
