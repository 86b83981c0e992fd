#!/usr/bin/env python3
"""PFST train-step throughput on MI355X (BASELINE.json metric) + roofline of the dominant kernel + CPU baseline.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...          (no WORLD_SIZE in the environment: starts exactly that torch.distributed.run command
                                         as a fresh child process BEFORE any GPU call and relays rank 0's JSON line;
                                         --dry-run prints the child command instead -- what tools/dist_train.sh:8-17 does)

One "step" = one full PFGST.train_step (student fwd/bwd on source + mixed batch, EMA-teacher forward, pseudo labels,
class mix, PFGSTLoss, backward, gradient all-reduce, AdamW) on a synthetic batch of 8 x 1024x1024x3 tiles per GPU that
is resident in HBM before the timed region.  value = global images / s (one image = one source+target pair).
Rank 0 prints ONE JSON line.  `value` is the product configuration (teacher pass and weight gradients on side streams); the
dominant kernel's `roofline` comes from a second timed region of the same K steps on ONE stream (`alt_single_stream`), because a
kernel's duration is only defined while nothing else shares the CUs.  `per_rank` / `rank_spread`: step time min / mean / max,
the reducer's buckets, the all-reduce time the backward sweep did not hide, the host's wait in the step's one blocking read."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOAD = 'pfst_pots_irrg2vaih_irrg_deeplabv3plus_r50-d8'
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec
PEAK_HBM_GBPS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA, spec (a tuned loop on random data sustains 1250-1500: DVFS)
# expected dominant kernel per arithmetic (checked against the full per-kernel pass); keys are the timer's names: C-ABI entry + row tile
DOMINANT_KERNEL = {'f32': 'conv_igemm_q_kernel<128>', 'bf16x6': 'conv_igemm_split_kernel<128>', 'f16x3': 'conv_igemm_f16x3_kernel'}
MATH_DTYPE = {'f32': 'f32 (fp32-input MFMA v_mfma_f32_32x32x2_f32, fp32 accumulate)',
              'bf16x6': 'f32 (bf16x6 split MFMA, fp32 accumulate)',
              'f16x3': 'f32 (f16x3 split MFMA: two scaled fp16 pieces per operand, three products, fp32 accumulate)'}


class KernelTimer:
    """Brackets every C-ABI launch with HIP events on the launch stream (torch's current stream IS the stream the
    kernels are enqueued on) and books algorithmic flops for the MFMA convolution kernels."""

    def __init__(self, inner):
        self.inner = inner
        self.records = []      # (key, start_evt, end_evt, flops, bytes, fused-epilogue operand bytes)
        self.fused_bytes = {}
        self.enabled = False
        self.per_layer = False
        self.only = None       # kernel-name prefix: bracket only those launches (the timed region), None = all

    @staticmethod
    def _conv_variant(name, a):
        if name == 'pfst_conv_igemm':
            n, c, hi, wi, m, ho, wo, ks, mode = a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[17]
            bm = 128 if m > 64 else (64 if m > 32 else 32)
            px = ho * wo if mode == 0 else hi * wi            # forward-output pixels = algorithmic work
            nbytes = 4.0 * (n * c * hi * wi + n * m * ho * wo * (2 if a[18] else 1) + c * ks * ks * m)
            kern = f'conv_igemm_q_kernel<{bm}>' if c % 16 == 0 else f'conv_igemm_kernel<{bm}>'   # dispatch of pfst_conv_igemm
            return kern, 2.0 * n * m * c * ks * ks * px, nbytes
        if name == 'pfst_conv_wgrad':
            n, ci, co, ho, wo, ks, stride, dil = a[5], a[6], a[9], a[10], a[11], a[12], a[13], a[14]
            bm = 128 if co > 64 else (64 if co > 32 else 32)
            nbytes = 4.0 * (n * ci * a[7] * a[8] + n * co * ho * wo + 2 * co * ci * ks * ks)
            quad = stride == 1 and ((ks == 1 and (ho * wo) % 4 == 0) or (ks == 3 and wo % 16 == 0 and dil <= 8))   # pfst_wgrad_q_eligible
            kern = f'conv_wgrad_q_kernel<{bm},{ks * ks}>' if quad else f'conv_wgrad_kernel<{bm},{ks * ks}>'
            return kern, 2.0 * n * co * ci * ks * ks * ho * wo, nbytes
        if name == 'pfst_conv_igemm_split':
            n, c, hi, wi, m, ho, wo, ks, mode = a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[17]
            bm = 128 if m > 64 else (64 if m > 32 else 32)
            px = ho * wo if mode == 0 else hi * wi
            return f'conv_igemm_split_kernel<{bm}>', 2.0 * n * m * c * ks * ks * px, 4.0 * (n * c * hi * wi + n * m * ho * wo) + 6.0 * c * ks * ks * m
        if name == 'pfst_conv_igemm_f16x3':
            n, c, hi, wi, m, ho, wo, ks, mode = a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[19]
            px = ho * wo if mode == 0 else hi * wi
            kern = 'conv_igemm_f16x3_kernel' if m > 64 else 'conv_igemm_f16x3_kernel<64>'      # the 64-row tile is a family of its own, like split<64>
            return kern, 2.0 * n * m * c * ks * ks * px, 4.0 * (n * c * hi * wi + n * m * ho * wo) + 4.0 * c * ks * ks * m
        if name == 'pfst_wino_gemm_f16x3':
            n, k, m, t, nx = a[5], a[6], a[7], a[8], (a[9] + 2) ** 2
            return 'conv_igemm_f16x3_kernel', 2.0 * nx * n * m * k * t, 4.0 * nx * (n * k * t + n * m * t) + 4.0 * nx * k * m
        if name == 'pfst_conv_wgrad_f16x3':
            n, ci, co, hw = a[5], a[6], a[7], a[8]
            return 'conv_wgrad_f16x3_kernel', 2.0 * n * co * ci * hw, 4.0 * (n * ci * hw + n * co * hw + 2 * co * ci)
        if name == 'pfst_conv_wgrad_f16x3_q':
            n, ci, h, w_, co, ks = a[5], a[6], a[7], a[8], a[9], a[10]
            return 'conv_wgrad_q16_kernel', 2.0 * n * co * ci * ks * ks * h * w_, 4.0 * (n * ci * h * w_ + n * co * h * w_ + 2 * co * ci * ks * ks)
        if name == 'pfst_absmax':
            return name, 0.0, 4.0 * a[1] * a[2]
        if name == 'pfst_wino_gemm_split':
            n, k, m, t, nx = a[3], a[4], a[5], a[6], (a[7] + 2) ** 2
            bm = 128 if m > 64 else (64 if m > 32 else 32)
            return f'conv_igemm_split_kernel<{bm}>', 2.0 * nx * n * m * k * t, 4.0 * nx * (n * k * t + n * m * t) + 6.0 * nx * k * m
        if name == 'pfst_conv_wgrad_split':
            n, ci, co, ho, wo, ks = a[5], a[6], a[9], a[10], a[11], a[12]
            bm = 128 if co > 64 else (64 if co > 32 else 32)
            return f'conv_wgrad_split_q_kernel<{bm}>', 2.0 * n * co * ci * ks * ks * ho * wo, 4.0 * (n * ci * a[7] * a[8] + n * co * ho * wo + 2 * co * ci * ks * ks)
        # Winograd path: the X = (m+2)^2 transform-domain GEMMs are one launch of the same K-quad kernel (gridDim.y = X); the
        # flops booked are the GEMM's own (what the kernel executes), not the direct-convolution count it replaces
        if name == 'pfst_wino_gemm':
            n, k, m, t, nx = a[3], a[4], a[5], a[6], (a[7] + 2) ** 2
            bm = 128 if m > 64 else (64 if m > 32 else 32)
            return f'conv_igemm_q_kernel<{bm}>', 2.0 * nx * n * m * k * t, 4.0 * nx * (n * k * t + n * m * t + k * m)
        if name == 'pfst_wino_wgrad':
            n, ci, co, t, nx = a[4], a[5], a[6], a[7], (a[8] + 2) ** 2
            bm = 128 if co > 64 else (64 if co > 32 else 32)
            kern = 'conv_wgrad_f16x3_kernel' if (a[9] == 2 and co > 64) else f'conv_wgrad_split_q_kernel<{bm}>' if a[9] else f'conv_wgrad_q_kernel<{bm},1>'
            return kern, 2.0 * nx * n * co * ci * t, 4.0 * nx * (n * ci * t + n * co * t + 2 * co * ci)
        if name in ('pfst_wino_input', 'pfst_wino_dy'):          # read the image once, write X transform planes of T = HW/m^2 tiles
            return name, 0.0, 4.0 * a[3] * a[4] * a[5] * a[6] * (1.0 + (a[8] + 2) ** 2 / a[8] ** 2)
        if name == 'pfst_wino_output':
            return name, 0.0, 4.0 * a[3] * a[4] * a[5] * a[6] * ((a[15] + 2) ** 2 / a[15] ** 2 + (2 if a[8] else 1) + (1 if a[11] else 0))
        # HBM-bound kernels: read-once / write-once algorithmic bytes, fp32
        if name == 'pfst_dwconv3x3':
            n, c, h, w_, acc = a[5], a[6], a[7], a[8], a[11]
            return name, 0.0, 4.0 * n * c * h * w_ * (3 if acc else 2) + 36.0 * c
        if name == 'pfst_dwconv3x3_bwd':                  # both gradients in one pass: read dy, x (+ the pre-BN tensor when BN backward is folded in), write dx
            return name, 0.0, 4.0 * a[8] * a[9] * a[10] * a[11] * (3 + (1 if a[13] else 0) + (1 if a[15] else 0))
        if name == 'pfst_dwconv3x3_multi_fwd':            # ASPP: read x once, write the ns branch outputs
            return name, 0.0, 4.0 * a[10] * a[11] * a[12] * a[13] * (1 + a[2])
        if name == 'pfst_dwconv3x3_multi_bwd':            # read x and ns x (dy [+ pre-BN tensor]), write (read-modify-write) dx
            return name, 0.0, 4.0 * a[14] * a[15] * a[16] * a[17] * (2 + (1 if a[11] else 0) + a[2] * (2 if a[12] else 1))
        if name == 'pfst_bn_backward_sums':               # the reduction pass alone (no fused partials): read dy, x
            return name, 0.0, 0.0 if a[14] else 4.0 * a[10] * a[11] * a[12] * 2
        if name == 'pfst_dwconv3x3_wgrad':
            return name, 0.0, 4.0 * a[5] * a[6] * a[7] * a[8] * 2
        if name == 'pfst_bn_stats':
            return name, 0.0, 4.0 * a[2] * a[3] * a[4]
        if name == 'pfst_bn_apply':                       # read x (+ residual), write y
            return name, 0.0, 4.0 * a[10] * a[11] * a[12] * (3 if a[2] else 2)
        if name == 'pfst_bn_backward':                    # minimum: read dy, x (+ y for the residual form), write dx (+ dres)
            nel = 4.0 * a[17] * a[18] * a[19]
            return name, 0.0, nel * (3 + (1 if a[2] else 0) + (1 if a[12] else 0))
        if name == 'pfst_bn_backward_dual':               # two layers behind one gated gradient: read dy, xa, xb, write dxa, dxb (booked with the family)
            return 'pfst_bn_backward', 0.0, 4.0 * a[27] * a[28] * a[29] * 5
        if name == 'pfst_sim_map':                        # read the feature map once, write the 9-channel map
            return name, 0.0, 4.0 * a[1] * a[3] * a[4] * (a[2] + 10)
        if name == 'pfst_sim_map_bwd':                    # read features + 9-channel maps, write the feature gradient
            return name, 0.0, 4.0 * a[4] * a[6] * a[7] * (2 * a[5] + 19)
        return name, 0.0, 0.0

    def call(self, name, *args):
        if not self.enabled:
            return self.inner(name, *args)
        key, flops, nbytes = self._conv_variant(name, args)
        if self.only is not None and not key.startswith(self.only):
            return self.inner(name, *args)
        if self.per_layer and flops > 0:
            idx = {'pfst_conv_igemm': (6, 7, 8, 10, 13, 15, 17, 18), 'pfst_conv_wgrad': (5, 6, 7, 9, 12, 14),
                   'pfst_wino_gemm': (3, 4, 5, 6), 'pfst_wino_wgrad': (4, 5, 6, 7),
                   'pfst_conv_igemm_split': (6, 7, 8, 10, 13, 15, 17, 18), 'pfst_conv_wgrad_split': (5, 6, 7, 9, 12, 14),
                   'pfst_wino_gemm_split': (3, 4, 5, 6), 'pfst_conv_igemm_f16x3': (8, 9, 10, 12, 15, 17, 19, 20),
                   'pfst_wino_gemm_f16x3': (5, 6, 7, 8), 'pfst_conv_wgrad_f16x3': (5, 6, 7, 8),
                   'pfst_conv_wgrad_f16x3_q': (5, 6, 7, 9, 10, 11)}[name]
            key = key + (' wino ' if 'wino' in name else ' ') + ' '.join(str(args[i]) for i in idx)
        extra = 0.0
        if name == 'pfst_conv_igemm_f16x3':
            # operands of the fused epilogues, NOT part of the convolution's algorithmic bytes but part of the launch's HBM traffic: the pre-BN
            # tensor a data-gradient launch reads for the BatchNorm-backward sums it emits (bnb), the gated identity gradient it adds
            # (gate_dy), the old contents of an accumulated output -- one [N][M][pixels] fp32 tensor each
            a = args
            px = a[13] * a[14] if a[19] == 0 else a[10] * a[11]
            extra = 4.0 * a[8] * a[12] * px * (bool(a[22]) + bool(a[23]) + bool(a[20]))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        self.inner(name, *args)
        e.record()
        self.records.append((key, s, e, flops, nbytes, extra))

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        self.fused_bytes = {}          # per key: bytes of the fused epilogues' extra operands over the same launches
        for key, s, e, flops, nbytes, extra in self.records:
            d = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += s.elapsed_time(e)
            d[2] += flops
            d[3] += nbytes
            self.fused_bytes[key] = self.fused_bytes.get(key, 0.0) + extra
        return agg


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_pmc_hbm_traffic_per_launch.json, newest round: FETCH_SIZE and WRITE_SIZE collected in separate runs, KiB units,
    FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md).  None if no record matches."""
    import glob
    tag = 'f16x3' if 'f16x3' in kernel else 'bf16x6' if 'split' in kernel else 'f32'
    paths = sorted(p for p in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic_per_launch*.json'))
                   if ('f16x3' if 'f16x3' in os.path.basename(p) else 'bf16x6' if 'bf16x6' in os.path.basename(p) else 'f32') == tag)
    if not paths:
        return None, None
    rec = json.load(open(paths[-1]))        # the newest round's record of that arithmetic
    source = 'committed PMC pass of this command, NOT this run: profiles/' + os.path.basename(paths[-1])
    def split(k):              # 'conv_igemm_q_kernel<128, 0>' -> ('conv_igemm_q_kernel', ['128', '0'])
        k = k.replace(' ', '')
        if '<' not in k:
            return k, []
        base, args = k.split('<', 1)
        return base, args.rstrip('>').split(',')
    base, want = split(kernel)
    # the timer names a kernel family by C-ABI entry + row tile; on the device the 128-row split GEMMs are several kernels (the
    # software-pipelined K=16 loop, the K=32 pairing and its fused-epilogue variants, the plain loop for short contractions)
    family = {'conv_igemm_split_kernel': ('conv_igemm_split_pair_kernel', 'conv_igemm_split_pair_bnb_kernel', 'conv_igemm_split_pipe_kernel'),
              'conv_wgrad_split_q_kernel': ('conv_wgrad_split_q_pipe_kernel',)}.get(base, ()) if want == ['128'] else ()
    if base == 'conv_igemm_f16x3_kernel':          # + the variants with the fused BatchNorm-backward epilogue and the normalise-on-load form
        family = ('conv_igemm_f16x3_bnb_kernel', 'conv_igemm_f16x3_bnl_kernel')
    if base == 'conv_wgrad_f16x3_kernel':          # the K=32-per-step kernels the entry point launches by default (whole-line loads)
        family = ('conv_wgrad_f16x3_pair_kernel', 'conv_wgrad_f16x3_line_kernel')
    tot, calls = 0.0, 0
    for k, v in rec.items():          # all instantiations whose leading template arguments match (e.g. the fused-epilogue variants
        if not isinstance(v, dict):   # <128, 0..3> of conv_igemm_q_kernel<128>), weighted by their launch counts
            continue
        b, a = split(k)
        if base == 'conv_igemm_f16x3_kernel' and not want and a and a[-1] == '64':
            continue                  # the 64-row tile is timed as a family of its own ('conv_igemm_f16x3_kernel<64>')
        if (b == base and a[:len(want)] == want) or b in family:
            tot += (v['fetch_MB_corrected'] + v['write_MB']) * 1e6 * v['calls']
            calls += v['calls']
    return (tot / calls, source) if calls else (None, None)


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(num_classes, threads, timed_steps=3):
    """The oracle (CPU restatement of the reference path) timed on this host as SURVEY.md §8d / BASELINE.md §3 specify:
    1 warm-up + 3 timed PFGST.train_step on a bounded sample (b=2, 512x512 crops = 1/4 of a 1024^2 tile each; the full
    b=8 x 1024^2 step would take minutes), torch.set_num_threads(host core share), reported in 1024^2-tile-equivalent
    images/s with the spread of the three steps."""
    from oracle import pfst_oracle as O
    from pfst_amd.synthetic import fill_state_dict, synth_batch
    torch.set_num_threads(threads)
    sd = fill_state_dict(O.init_state_dict(num_classes, 3), 0)
    m = O.OraclePFGST(sd)
    b, S = 2, 512
    batch = synth_batch(b, S, num_classes, seed=1234)
    m.train_step(batch)                       # warm-up (allocator, thread pool, oneDNN primitive caches)
    dts = []
    for _ in range(timed_steps):
        t0 = time.perf_counter()
        m.train_step(batch)
        dts.append(time.perf_counter() - t0)
    tiles = b * (S * S) / (1024.0 * 1024.0)
    mean = sum(dts) / len(dts)
    spread = round((max(dts) - min(dts)) / mean, 3)
    return dict(value=tiles / mean, unit='images/s', cores=threads, kind='port', cpu_model=_cpu_model(),
                step_seconds=[round(d, 2) for d in dts], spread=spread,
                note=f'spread of the timed steps above 10 %: shared host, treat the value as +-{round(100 * spread)} %' if spread > 0.1 else None,
                sample=f'{timed_steps} steps after 1 warm-up: PFGST.train_step, b={b}, {S}x{S} crops (={tiles:.2f} 1024^2-tile '
                       f'equivalents per step), mean {mean:.1f} s/step, torch-CPU fp32 oracle on {threads} threads')


def launch_ranks(args, argv):
    """`bench.py --gpus N` without a torch.distributed.run environment: start N ranks (one process per GPU, RCCL over xGMI) the way
    the reference's launcher does (tools/dist_train.sh:8-17: `python -m torch.distributed.launch --nproc_per_node=$GPUS
    --master_port=$PORT train.py --launcher pytorch`).  This process has made no GPU call (importing torch makes none), starts the
    launcher as a CHILD -- never an exec: a process that has touched the GPU must not be replaced -- relays its output and exits with
    its code.  The port is taken from --master-port / MASTER_PORT or picked free on 127.0.0.1."""
    import socket
    import subprocess
    port = args.master_port or int(os.environ.get('MASTER_PORT', 0))
    if not port:
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    child_args = [a for a in argv if a != '--dry-run']
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + child_args
    if args.dry_run:
        print(json.dumps({'launcher': 'child process', 'cmd': cmd}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the config\'s 8)')
    ap.add_argument('--size', type=int, default=None, help='tile size (default 1024)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--per-layer', action='store_true', help='debug: per-layer conv timing table on stderr')
    ap.add_argument('--no-alt-math', action='store_true', help='skip the informational passes: the other two arithmetics (N=1 only)')
    ap.add_argument('--math', choices=['f32', 'bf16x6', 'f16x3'], default=None, help='arithmetic of the dense convolutions for `value` '
                    '(default: the product default, pfst_amd.layers.CONV_MATH / PFST_CONV_MATH)')
    ap.add_argument('--master-port', type=int, default=None, help='rendezvous port when bench.py starts the ranks itself')
    ap.add_argument('--rendezvous-only', action='store_true', help='launch-path check: join the process group, one all-reduce, no kernels')
    ap.add_argument('--dry-run', action='store_true', help='with --gpus N > 1 and no WORLD_SIZE: print the child command and exit')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    if args.dry_run:
        print(json.dumps({'launcher': 'none (single process)' if args.gpus == 1 else 'torch.distributed.run environment present',
                          'cmd': [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != '--dry-run']}), flush=True)
        return

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.rendezvous_only:
        # launch-path check (no kernels): every rank joins the group the bench would use -- RCCL when each rank has its own GPU,
        # gloo otherwise (the CPU test of the launcher) -- and takes part in one all-reduce; rank 0 prints the group it saw
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        use_rccl = torch.cuda.device_count() >= world and torch.cuda.is_available()
        if world > 1:
            if use_rccl:
                torch.cuda.set_device(local_rank)
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
            else:
                dist.init_process_group('gloo', rank=rank, world_size=world)
            t = torch.ones(4, device=torch.device('cuda', local_rank) if use_rccl else 'cpu') * (rank + 1)
            dist.all_reduce(t)
            assert float(t[0]) == world * (world + 1) / 2
        if rank == 0:
            print(json.dumps({'rendezvous': 'ok', 'n_gpus': args.gpus, 'rccl_ranks': dist.get_world_size() if world > 1 else 1,
                              'backend': dist.get_backend() if world > 1 else None}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        assert args.gpus == world, f'--gpus {args.gpus} but WORLD_SIZE={world}'
        return
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
    assert args.gpus == world, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    import pfst_amd  # noqa: F401
    from pfst_amd import dist as pdist
    from pfst_amd import hip_ops, layers, strong_aug  # noqa: F401
    from pfst_amd.hostinfo import usable_cpus
    from pfst_amd.optim import build_optimizer, poly_lr
    from pfst_amd.presets import OPTIMIZER, workload_cfg
    from pfst_amd.registry import UDA
    from pfst_amd.synthetic import fill_state_dict, synth_batch

    cfg, w = workload_cfg(WORKLOAD)
    b = args.batch or w['per_gpu_batch']
    S = args.size or w['size']
    main_math = args.math or layers.CONV_MATH          # the arithmetic `value` is measured in (the product default)
    other_math = 'f32' if main_math != 'f32' else 'bf16x6'     # the line always carries the fp32-input MFMA step beside a split arithmetic
    batch = synth_batch(b, S, w['num_classes'], w['in_channels'], seed=1234 + rank, device=dev)

    timer = KernelTimer(hip_ops.call)
    state = dict(it=0)

    def build(math):
        layers.CONV_MATH = math
        model = UDA.build(cfg)
        fill_state_dict(model.state_dict(), 0)            # same seeded weights on every rank (DDP starts in sync)
        model.to(dev)
        return model, build_optimizer(model, OPTIMIZER)

    def run_steps(model, opt, n, marks=None):
        for _ in range(n):
            for g in opt.param_groups:
                g['lr'] = poly_lr(OPTIMIZER['lr'], state['it'], cfg['max_iters'])
            out = model.train_step(batch, opt)
            state['it'] += 1
            if marks is not None:
                # DEVICE clock: an event behind the step's last launch (the optimizer kernel).  The host clock no longer brackets a step: its
                # blocking read waits for the step's FORWARD passes only (layers.EARLY_LOG_READ), the host runs up to one backward sweep ahead
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append(e)
        return out

    def timed_steps(model, opt, n, times=None):
        """-> (wall seconds of the n steps between two synchronisations, the last step's outputs); times: filled with the n per-step durations
        on the device (seconds, event to event on the launch stream)"""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        marks = None
        if times is not None:
            marks = [torch.cuda.Event(enable_timing=True)]
            marks[0].record()
        t0 = time.perf_counter()
        out = run_steps(model, opt, n, marks)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if times is not None:
            times.extend(1e-3 * marks[i].elapsed_time(marks[i + 1]) for i in range(n))
        return dt, out

    def kernel_tables(model, opt, math, dom_agg):
        """second, untimed pass of the same K steps with EVERY launch bracketed -> roofline legs and per-kernel tables"""
        hip_ops.call = timer.call
        timer.records, timer.only, timer.enabled, timer.per_layer = [], None, True, args.per_layer
        run_steps(model, opt, args.steps)
        torch.cuda.synchronize()
        timer.enabled = False
        hip_ops.call = timer.inner
        agg = timer.summary()
        timer.records = []
        out = {}
        if args.per_layer:       # debug table, then fold back to per-kernel keys for the JSON line
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                if v[2] == 0:
                    sys.stderr.write(f'{k:70s} calls/step {v[0] / args.steps:5.1f}  ms/step {v[1] / args.steps:7.3f}\n')
                if v[2] > 0:
                    sys.stderr.write(f'{k:70s} calls/step {v[0] / args.steps:5.1f}  ms/call {v[1] / v[0]:7.3f}  TF/s {v[2] / v[1] / 1e9:6.1f}\n')
            folded = {}
            for k, v in agg.items():
                d = folded.setdefault(k.split(' ')[0] if v[2] > 0 else k, [0, 0.0, 0.0, 0.0])
                for i in range(4):
                    d[i] += v[i]
            agg = folded
        tot_ms = sum(v[1] for v in agg.values())
        mfma = {k: v for k, v in agg.items() if v[2] > 0}
        split = math != 'f32'
        # dominant kernel = the MFMA kernel with the most time; a split implicit GEMM is priced against the bf16 / fp16 dense peak with
        # the MFMA flops it EXECUTES per algorithmic flop (6 for bf16x6, 3 for f16x3)
        fam = {'f32': lambda k: 'split' not in k and 'f16x3' not in k, 'bf16x6': lambda k: 'split' in k, 'f16x3': lambda k: 'f16x3' in k}[math]
        cand = {k: v for k, v in mfma.items() if fam(k)} or mfma
        dom = max(cand.items(), key=lambda kv: kv[1][1])
        measured_in = 'table pass (single stream, all launches bracketed)'
        if dom_agg and dom[0] in dom_agg:          # the expected dominant kernel: its launches inside the TIMED region
            dom = (dom[0], dom_agg[dom[0]])
            measured_in = 'timed single-stream region of this run (the K steps after the `value` region, one stream: a kernel duration needs the device to itself)'
        cnt, ms, fl, nb = dom[1]
        # roofline.achieved = ALGORITHMIC flops per launch / the kernel's average launch duration, against the dense peak of the
        # matrix pipe the kernel runs on.  A split kernel executes `mult` MFMA flops per algorithmic flop (3 for f16x3, 6 for bf16x6):
        # that product is matrix-pipe UTILISATION -- `mfma_pipe_utilisation`, reported beside the roofline, never as it.
        mult, peak, unit = (3.0, PEAK_BF16_MFMA_TFLOPS, 'TFLOP/s (algorithmic flops; fp16 MFMA dense peak)') if 'f16x3' in dom[0] else \
            (6.0, PEAK_BF16_MFMA_TFLOPS, 'TFLOP/s (algorithmic flops; bf16 MFMA dense peak)') if 'split' in dom[0] else \
            (1.0, PEAK_FP32_MFMA_TFLOPS, 'TFLOP/s')
        achieved = fl / (ms * 1e-3) / 1e12

        def pipe(mult_, alg_tflops):
            return {'mfma_flops_per_algorithmic_flop': mult_, 'executed_tflops': mult_ * alg_tflops, 'frac_of_peak': mult_ * alg_tflops / peak,
                    'structural_ceiling_tflops_algorithmic': peak / mult_, 'frac_of_structural_ceiling': alg_tflops / (peak / mult_)}
        traffic, src = pmc_traffic(dom[0])
        fused_per_launch = timer.fused_bytes.get(dom[0], 0.0) / agg[dom[0]][0] if dom[0] in agg else None      # (from the table pass: same launches)
        out['roofline'] = {'kernel': dom[0], 'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': unit, 'frac': achieved / peak,
                           'traffic': traffic, 'traffic_source': src, 'launches': cnt, 'avg_launch_ms': ms / cnt, 'measured_in': measured_in,
                           'algorithmic_flops_per_launch': fl / cnt, 'mfma_pipe_utilisation': pipe(mult, achieved),
                           'fp32_equivalent_tflops': achieved,
                           'fp32_equivalent_vs_fp32_mfma_peak': achieved / PEAK_FP32_MFMA_TFLOPS,
                           'algorithmic_bytes_per_launch': nb / cnt,
                           'fused_epilogue_operand_bytes_per_launch': fused_per_launch,
                           'traffic_note': 'traffic = L2-miss bytes by the counters, all launches of the family; algorithmic bytes = the convolution alone; the '
                                           'fused epilogues (BatchNorm-backward sums, gated identity gradient, accumulation) read further whole tensors, '
                                           'counted in fused_epilogue_operand_bytes_per_launch',
                           'ms_per_step': ms / args.steps}
        # the weight-gradient kernel in the same form, so its over-fetch ratio (PMC traffic vs algorithmic bytes) is visible too
        wk = 'conv_wgrad_f16x3_kernel' if math == 'f16x3' else 'conv_wgrad_split_q_kernel<128>' if split else 'conv_wgrad_q_kernel<128,1>'
        if wk in agg:
            wc, wms, wfl, wnb = agg[wk]
            wt, wsrc = pmc_traffic(wk)
            wmult = 3.0 if 'f16x3' in wk else 6.0 if 'split' in wk else 1.0
            wach = wfl / (wms * 1e-3) / 1e12
            out['roofline_wgrad'] = {'kernel': wk, 'bound': 'mfma', 'achieved': wach, 'peak': peak, 'unit': unit, 'frac': wach / peak,
                                     'traffic': wt, 'traffic_source': wsrc, 'launches': wc, 'avg_launch_ms': wms / wc,
                                     'measured_in': 'second pass (all launches bracketed)', 'fp32_equivalent_tflops': wach,
                                     'mfma_pipe_utilisation': pipe(wmult, wach),
                                     'algorithmic_flops_per_launch': wfl / wc, 'algorithmic_bytes_per_launch': wnb / wc}
        all_fl = sum(v[2] for v in mfma.values())
        all_ms = sum(v[1] for v in mfma.values())
        out['mfma_all_convs'] = {'fp32_equivalent_tflops': all_fl / (all_ms * 1e-3) / 1e12, 'ms_per_step': all_ms / args.steps,
                                 'share_of_kernel_time': all_ms / tot_ms}
        # the memory-bound kernels against the HBM roofline: algorithmic bytes (read-once / write-once) / kernel time.
        # bn_backward is two passes (reduce + apply) over a 3-tensor minimum, so its ceiling is 3/5 of the peak.
        out['hbm_kernels'] = {k: {'GBps': round(v[3] / (v[1] * 1e-3) / 1e9, 1), 'frac_of_8TBps': round(v[3] / (v[1] * 1e-3) / 8e12, 3),
                                  'ms_per_step': round(v[1] / args.steps, 3)}
                              for k, v in agg.items() if v[2] == 0 and v[3] > 0}
        out['kernel_ms_per_step'] = {k: round(v[1] / args.steps, 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]}
        out['kernel_time_total_ms_per_step'] = tot_ms / args.steps
        return out

    def measure(math, tables=True):
        """one arithmetic: warm-up; the timed K steps of the PRODUCT configuration (`value`: teacher pass forked, weight gradients on the
        side stream -- layers.FORK_TEACHER / WGRAD_STREAM, nothing bracketed); then the same K steps on ONE stream with the dominant
        kernel's launches bracketed by HIP events (a per-kernel duration is only defined while nothing co-runs on the CUs: `roofline`,
        and the single-stream step as `alt_single_stream`); then the table pass (every launch bracketed)."""
        model, opt = build(math)
        dominant = DOMINANT_KERNEL[math]
        product = (layers.WGRAD_STREAM, layers.FORK_TEACHER)
        hip_ops.call = timer.inner
        run_steps(model, opt, args.warmup)
        rec = {}
        pdist.STEP_STATS = []
        step_s = []
        elapsed, out = timed_steps(model, opt, args.steps, step_s)
        stats, pdist.STEP_STATS = pdist.STEP_STATS, None
        mine = dict(rank=rank, step_ms=dict(min=round(1e3 * min(step_s), 2), mean=round(1e3 * sum(step_s) / len(step_s), 2), max=round(1e3 * max(step_s), 2)),
                    **pdist.summarize_step_stats(stats))
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        else:
            per_rank = [mine]
        rec.update(value=b * world * args.steps / elapsed, unit='images/s', ms_per_step=1000.0 * elapsed / args.steps,
                   loss=out['log_vars'].get('decode.loss_ce'), per_rank=per_rank,
                   streams='teacher forward forked + weight gradients on a side stream' if any(product) else 'one stream')
        if world > 1:
            means = [r['step_ms']['mean'] for r in per_rank]
            rec['rank_spread'] = dict(step_ms_mean_min=min(means), step_ms_mean_max=max(means),
                                      slowest_rank=int(max(range(world), key=lambda i: means[i])),
                                      exposed_allreduce_ms_mean_max=max((r['exposed_allreduce_ms'] or {}).get('mean', 0.0) for r in per_rank))
        timing = not args.no_kernel_timing
        if timing:
            # Single-stream region: the same K steps, only the dominant kernel's launches bracketed with events (2700 event pairs per step
            # would cost the step 1.5-2 %, these 272 cost 1 ms).  Its roofline figure is an average over launches that own the device.
            layers.set_overlap(False, False)
            run_steps(model, opt, 1)
            if not args.per_layer:
                hip_ops.call = timer.call
                timer.records, timer.only, timer.enabled, timer.per_layer = [], dominant, True, False
            elapsed1, _ = timed_steps(model, opt, args.steps)
            timer.enabled = False
            hip_ops.call = timer.inner
            dom_agg = timer.summary() if not args.per_layer else None
            timer.records = []
            rec['single_stream'] = {'mode': 'the same step on one stream (PFST_WGRAD_STREAM=0 PFST_FORK_TEACHER=0), dominant kernel bracketed: the '
                                            'region `roofline` is measured in', 'value': b * args.steps / elapsed1, 'unit': 'images/s (this rank)',
                                    'ms_per_step': 1000.0 * elapsed1 / args.steps}
            if rank == 0 and tables:
                rec.update(kernel_tables(model, opt, math, dom_agg))
            elif tables:
                run_steps(model, opt, args.steps)          # keep the ranks in step (collectives inside train_step)
            layers.set_overlap(*product)
        rec['hbm_peak_allocated_GB'] = round(torch.cuda.max_memory_allocated(dev) / 1e9, 1)
        del model, opt
        torch.cuda.empty_cache()
        return rec

    main = measure(main_math)
    res = None
    if rank == 0:
        global_batch = b * world
        res = {
            'metric': 'PFST train-step images/s on 1024² IRRG tiles', 'value': main['value'], 'unit': 'images/s',
            'n_gpus': world, 'rccl_ranks': dist.get_world_size() if world > 1 else 1, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': main['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': MATH_DTYPE[main_math], 'data': 'synthetic',
            'config': {'workload': WORKLOAD, 'global_batch': global_batch, 'per_gpu_batch': b, 'tile': f'{S}x{S}x{w["in_channels"]}',
                       'num_classes': w['num_classes'], 'parallelism': f'dp{world}', 'weights': 'seeded random init',
                       'conv_math': main_math,
                       'strong_aug': 'colour-jitter p=0.8 + gaussian-blur p=0.5 (HIP kernels; kornia arithmetic restated, PARITY UNPINNED, <2 ms of the step)',
                       'dropout': 0.1},
            'loss': main['loss'], 'hbm_peak_allocated_GB': main['hbm_peak_allocated_GB'],
        }
        for k in ('roofline', 'roofline_wgrad', 'mfma_all_convs', 'hbm_kernels', 'kernel_ms_per_step', 'kernel_time_total_ms_per_step'):
            if k in main:
                res[k] = main[k]
        for k in ('per_rank', 'rank_spread', 'streams'):
            if k in main:
                res[k] = main[k]
        if 'single_stream' in main:
            res['alt_single_stream'] = main['single_stream']
    if world == 1 and not args.no_alt_math:
        # the other arithmetic on the same step, same line: fp32-input MFMA (v_mfma_f32_32x32x2_f32) when `value` runs the
        # fp32-faithful bf16x6 split, and vice versa -- with its own roofline leg
        alt = measure(other_math)
        alt.pop('per_rank', None)
        alt['mode'] = MATH_DTYPE[other_math] + '; PFST_CONV_MATH=' + other_math
        alt.pop('hbm_kernels', None)
        res['alt_math'] = alt
        # the third arithmetic, timed only (no kernel tables): the step under the other split
        third = ({'f32', 'bf16x6', 'f16x3'} - {main_math, other_math}).pop()
        m3, o3 = build(third)
        hip_ops.call = timer.inner
        run_steps(m3, o3, args.warmup)
        dt3, _ = timed_steps(m3, o3, args.steps)
        res['alt_math_' + third] = {'mode': MATH_DTYPE[third] + '; PFST_CONV_MATH=' + third, 'value': b * args.steps / dt3, 'unit': 'images/s',
                                    'ms_per_step': 1000.0 * dt3 / args.steps}
        del m3, o3
        torch.cuda.empty_cache()
    layers.CONV_MATH = main_math
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            res['cpu_baseline'] = cpu_baseline(w['num_classes'], usable_cpus())
        elif not args.no_cpu_baseline:
            res['cpu_baseline'] = None   # measured at N=1 only
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
