"""What would running the two student passes as ONE batch of 16 through the convolutions buy?

Source and mixed pass share weights and shapes, so every dense convolution of a student pass could be launched once on 16 images instead
of twice on 8 (with per-pass BatchNorm statistics): half the launches' ramp-up / tail, weights read once, half the weight gradients'
atomic epilogues.  Before building it (every BatchNorm kernel, the fused statistics and the fused BatchNorm-backward epilogues would need
an image-group dimension) this tool measures the ceiling: every distinct convolution of the network as the product dispatches it (f16x3 /
Winograd / bf16x6 / fp32-MFMA by shape, layers.Conv2dP), forward + data gradient + weight gradient, timed at N = 8 and N = 16 on the
b = 8 x 1024^2 shapes; the saving per step is  sum over layers of calls x (2 t(8) - t(16)).  The HBM-bound kernels (BatchNorm, transforms'
streaming parts, depthwise) move the same bytes either way and are not part of the ceiling.

    python tools/batch16_probe.py [--size 1024]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=1024)
    args = ap.parse_args()
    import pfst_amd  # noqa: F401
    from pfst_amd import hip_ops as ops
    from pfst_amd import layers
    from pfst_amd.engine import ParamArena
    from pfst_amd.presets import model_cfg
    from pfst_amd.registry import build_segmentor

    dev = torch.device('cuda')
    model = build_segmentor(model_cfg(6, 3, dropout=0.0)).to(dev)
    ParamArena(list(model.named_parameters()), dev, with_grad=True)
    model.repack_weights(need_dgrad=True)
    names = {id(m): n for n, m in model.named_modules()}

    # one forward at N = 2 on a quarter-size image records every convolution's input shape (scaled back up below)
    seen = []
    orig = layers.Conv2dP.fprop

    def rec(self, xd, *a, **kw):
        seen.append((self, tuple(xd.shape)))
        return orig(self, xd, *a, **kw)
    layers.Conv2dP.fprop = rec
    s4 = args.size // 4
    with torch.no_grad():
        model.encode_decode(torch.randn(2, 3, s4, s4, device=dev), None)
    layers.Conv2dP.fprop = orig
    uniq = {}
    for conv, shp in seen:
        if conv.depthwise or shp[2] * shp[3] == 1:
            continue
        key = (conv.cin, conv.cout, conv.k, conv.stride, conv.dilation, shp[2] * 4, shp[3] * 4)
        uniq.setdefault(key, [conv, 0, names[id(conv)]])[1] += 1

    total = {8: 0.0, 16: 0.0}
    saving = 0.0
    print(f'{"layer (first of its shape)":34s} {"cin":>5s} {"cout":>5s} k s d {"HxW":>9s} calls | per direction: t(8) t(16) ms -> saving per student pass pair')
    for key, (conv, calls, name) in sorted(uniq.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][5] * kv[0][6] * kv[0][2] ** 2):
        cin, cout, k, stride, dil, H, W = key
        row = []
        t = {}
        for n in (8, 16):
            x = torch.relu(torch.randn(n, cin, H, W, device=dev))
            xa = ops.absmax(x) if (conv.f16_f or conv.wino_f16 or layers.CONV_MATH == 'f16x3') else None
            y = conv.fprop(x, keep=True, x_amax=xa if (conv.f16_f or conv.wino_f16) else None)
            saved_v = conv.saved_v
            dy = torch.randn_like(y) * 1e-3
            da = ops.absmax(dy) if layers.CONV_MATH == 'f16x3' else None
            dx = torch.empty_like(x)
            need_d = cin >= 16                                   # the stem's first convolution has no data gradient
            tf = timeit(lambda: conv.fprop(x, keep=False, x_amax=xa if (conv.f16_f or conv.wino_f16) else None))
            td = timeit(lambda: conv.dgrad(dy, (H, W), dx, False, dy_amax=da)) if need_d else 0.0
            tw = timeit(lambda: layers._wgrad(conv, x, dy, saved_v, xa, da))
            t[n] = (tf, td, tw)
            del x, y, dy, dx, saved_v
            torch.cuda.empty_cache()
        # per step: 2 student passes forward + backward (the teacher's forward stays a batch of 8)
        per8 = sum(t[8]) * calls * 2
        per16 = sum(t[16]) * calls
        total[8] += per8
        total[16] += per16
        saving += per8 - per16
        print(f'{name:34s} {cin:5d} {cout:5d} {k} {stride} {dil} {H:4d}x{W:<4d} {calls:5d} | fprop {t[8][0]:.3f} {t[16][0]:.3f}  dgrad {t[8][1]:.3f} {t[16][1]:.3f}  '
              f'wgrad {t[8][2]:.3f} {t[16][2]:.3f} -> {per8 - per16:+.3f} ms', flush=True)
    print(f'\nstudent passes, dense convolutions: 2 x N=8 {total[8]:.1f} ms per step, 1 x N=16 {total[16]:.1f} ms per step: ceiling of the batch-of-16 '
          f'schedule {saving:.1f} ms per step ({100 * saving / total[8]:.1f} % of these kernels)')


if __name__ == '__main__':
    main()
