#!/usr/bin/env python3
"""What would independent passes of the network gain from running on their own streams?  (A ceiling for a forked student schedule: the
source pass and the mixed pass share nothing but the weights until their gradients meet in the arena.)

  forward: P no-grad forward passes of the student (encode_decode, batch-statistics BatchNorm) -- one stream vs one stream each;
  pass:    P full student passes (forward_train + backward of that pass) -- one stream vs one stream each.  The concurrent passes race on
           the BatchNorm running statistics and on non-atomic gradient accumulations: TIMING ONLY, the gradients are not used.

  python tools/chain_overlap_probe.py --passes 2 --reps 3"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--passes', type=int, default=2)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--wgrad-stream', type=int, default=1)
    args = ap.parse_args()
    import bench
    from pfst_amd import hip_ops as ops, layers
    from pfst_amd.engine import Tape
    from pfst_amd.presets import workload_cfg
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import fill_state_dict, synth_batch

    dev = torch.device('cuda', 0)
    cfg, w = workload_cfg(bench.WORKLOAD)
    b, S = w['per_gpu_batch'], w['size']
    batch = synth_batch(b, S, w['num_classes'], w['in_channels'], seed=1234, device=dev)
    uda = UDA.build(cfg)
    fill_state_dict(uda.state_dict(), 0)
    uda.to(dev)
    uda._ensure_arenas(dev)
    model = uda.get_model()
    arena = uda._student_arena
    for name, p in model.named_parameters():
        p.grad = arena.view(arena.grad, name)
    model.repack_weights(need_dgrad=True)
    layers.set_overlap(bool(args.wgrad_stream), False)
    img = batch['img'].contiguous()
    gt8 = ops.to_u8(batch['gt_semantic_seg'].contiguous())
    metas = batch['img_metas']
    streams = [torch.cuda.Stream() for _ in range(args.passes)]

    def fwd():
        model.encode_decode(img, metas)

    def full():
        tape = Tape()
        model.forward_train(img, metas, gt8, None, tape=tape)
        tape.backward()

    def run(fn, concurrent):
        torch.cuda.synchronize()
        main_s = torch.cuda.current_stream()
        t0 = time.perf_counter()
        if concurrent:
            for s in streams:
                s.wait_stream(main_s)
                with torch.cuda.stream(s):
                    fn()
            for s in streams:
                main_s.wait_stream(s)
        else:
            for _ in streams:
                fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0), 1e3 * t_host

    for name, fn in (('forward', fwd), ('pass', full)):
        for conc in (False, True):                 # warm both forms (per-stream workspaces, allocator pools)
            run(fn, conc)
        for r in range(args.reps):
            ts, hs = run(fn, False)
            tc, hc = run(fn, True)
            print(f'{name:8s} x{args.passes}  one stream {ts:8.2f} ms (host {hs:6.1f})   own streams {tc:8.2f} ms (host {hc:6.1f})   '
                  f'{tc - ts:+7.2f} ms  {100 * (tc / ts - 1):+5.1f} %', flush=True)


if __name__ == '__main__':
    main()
