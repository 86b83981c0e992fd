"""Timing ablation of the bf16x6 implicit-GEMM main loop (PFST_SPLIT_DIAG, see conv_split.hip: 1-6 ablate / stamp the un-pipelined loop (8),
7 swaps the MFMA shape in the K=16 pipelined loop (9); 0 is what the product launches): which part of the loop the time
goes to.  The ablated variants compute wrong results on purpose; this tool only times them.  Run on the GPU box:
    for D in 0 8 1 2 3 4 5 6 9 7; do PFST_SPLIT_DIAG=$D python tools/split_ablation.py; done
(the tool sets PFST_DIAG_WRONG_RESULTS_OK=1 itself: the library refuses PFST_SPLIT_DIAG without it)"""
import os, sys
os.environ.setdefault('PFST_DIAG_WRONG_RESULTS_OK', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfst_amd import hip_ops as ops

NAMES = {0: 'product dispatch (K=32 pairing)', 8: 'un-pipelined loop', 9: 'K=16 pipelined loop', 1: 'no split+LDS store', 2: 'no global loads', 3: 'no MFMA', 4: 'no LDS fragment reads', 5: 'no in-loop barrier', 6: 'phase stamps', 7: 'pipelined, 16x16x32 MFMA shape'}


def timeit(fn, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); t.append(s.elapsed_time(e))
    return sorted(t)[len(t) // 2]


d = int(os.environ.get('PFST_SPLIT_DIAG', '0'))
row = []
for name, ci, co, k, dil, hin in [('l4.conv2', 512, 512, 3, 4, 128), ('aspp.pw', 2048, 512, 1, 1, 128), ('head.bottleneck', 2560, 512, 3, 1, 128)]:
    pad = dil if k == 3 else 0
    x = torch.randn(8, ci, hin, hin, device='cuda'); w = torch.randn(co, ci, k, k, device='cuda') * 0.05
    w6f, _ = ops.pack_weight_split(w)
    y = ops.conv_fprop_split(x, w6f, co, k, 1, dil, pad)
    t = timeit(lambda: ops.conv_fprop_split(x, w6f, co, k, 1, dil, pad, out=y))
    row.append(f'{name} {t:7.3f} ms')
    if d == 6:
        # per-wave sums of the four phase durations of a K-step (shader cycles): fragment reads landed | MFMAs issued | split + LDS store
        # done | barrier passed
        _, st, _ = ops.conv_fprop_split(x, w6f, co, k, 1, dil, pad, out=y, want_stats=True)
        torch.cuda.synchronize()
        waves = 8 * (hin * hin // 128) * (co // 128) * 4
        ph = st.view(torch.int32)[:waves * 8].view(-1, 8).double()
        kt = ci * k * k // 16
        m = ph.mean(0) / kt
        row.append('cycles per K-step: global loads issued %.0f | LDS fragments landed %.0f | MFMAs issued %.0f | global loads landed %.0f | '
                   'split + LDS store done %.0f | barrier passed %.0f | sum %.0f (768 = the MFMAs alone)'
                   % (m[4], m[0], m[1], m[5], m[2], m[3], m[:6].sum()))
print(f'diag {d} ({NAMES[d]:24s}): ' + ' | '.join(row), flush=True)
