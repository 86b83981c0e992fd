#!/usr/bin/env python3
"""What the deterministic mode (hip_ops.set_deterministic, `tools/train.py --deterministic`) costs at the bench shape: the same model, the
default and the fixed-order step alternately.   python tools/det_cost.py [--steps 4]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=4)
    args = ap.parse_args()
    import bench
    from pfst_amd import hip_ops
    from pfst_amd.optim import build_optimizer
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
    for _ in range(2):
        model.train_step(batch, opt)
    for rnd in range(2):
        for det in (False, True):
            hip_ops.set_deterministic(det)
            model.train_step(batch, opt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                model.train_step(batch, opt)
            torch.cuda.synchronize()
            ms = 1000.0 * (time.perf_counter() - t0) / args.steps
            print(f'round {rnd} deterministic={int(det)}  {ms:8.2f} ms/step  {w["per_gpu_batch"] * 1000.0 / ms:6.2f} images/s', flush=True)
    hip_ops.set_deterministic(False)


if __name__ == '__main__':
    main()
