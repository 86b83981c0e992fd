#!/usr/bin/env python3
"""Deterministic mode at the bench shape (b = 8 x 1024^2, stream overlap on): the same first train step from the same state twice -- the gradient
arenas must be bit-identical; lists the tensors that are not.  python tools/det_repro_fullsize.py [--runs 3]"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--runs', type=int, default=3)
    ap.add_argument('--last-single-stream', action='store_true', help='the last run on one stream: the step must not depend on the stream schedule either')
    args = ap.parse_args()
    import bench
    from pfst_amd import hip_ops, layers
    from pfst_amd.optim import build_optimizer
    from pfst_amd.presets import OPTIMIZER, workload_cfg
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import fill_state_dict, synth_batch
    dev = torch.device('cuda', 0)
    cfg, w = workload_cfg(bench.WORKLOAD)
    batch = synth_batch(w['per_gpu_batch'], w['size'], w['num_classes'], w['in_channels'], seed=1234, device=dev)
    hip_ops.set_deterministic(True)
    print('stream overlap:', layers.WGRAD_STREAM, layers.FORK_TEACHER)
    grads = []
    for r in range(args.runs):
        if args.last_single_stream and r == args.runs - 1:
            layers.set_overlap(False, False)
            print('last run: one stream')
        model = UDA.build(cfg)
        fill_state_dict(model.state_dict(), 0)
        model.to(dev)
        opt = build_optimizer(model, OPTIMIZER)
        random.seed(0); np.random.seed(0); torch.manual_seed(0); torch.cuda.manual_seed_all(0)
        out = model.train_step(batch, opt)
        torch.cuda.synchronize()
        a = model.student_arena
        grads.append(a.grad.clone())
        if r:
            layout = [(n, a.offsets[n], int(np.prod(a.shapes[n]))) for n in a.names]
            diff = [(n, int((grads[0][o:o + k] != grads[r][o:o + k]).sum()), float((grads[0][o:o + k] - grads[r][o:o + k]).norm() / grads[0][o:o + k].norm()))
                    for n, o, k in layout if not torch.equal(grads[0][o:o + k], grads[r][o:o + k])]
            print(f'run {r} vs run 0: {len(diff)} of {len(layout)} gradient tensors differ', diff[:8])
        del model, opt
        torch.cuda.empty_cache()
    hip_ops.set_deterministic(False)


if __name__ == '__main__':
    main()
