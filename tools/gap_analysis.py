"""Idle time between kernels on the GPU from a rocprofv3 --kernel-trace CSV (one process, default stream): for the dispatches of the LAST
`--steps` train steps of `bench.py --no-kernel-timing`, the wall span, the sum of kernel durations and the gaps between consecutive kernels.
    python tools/gap_analysis.py <dir with *kernel_trace.csv> [n_tail_kernels]"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
rows = rows[-n:]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
gaps = []
after = defaultdict(lambda: [0, 0])
cur_end = rows[0][1]
for i in range(1, len(rows)):
    g = rows[i][0] - cur_end
    if g > 0:
        gaps.append(g)
        k = rows[i - 1][2].split('(')[0][-40:]
        after[k][0] += 1
        after[k][1] += g
    cur_end = max(cur_end, rows[i][1])
idle = sum(gaps)
print(f'kernels {len(rows)}  span {span / 1e6:.2f} ms  sum of durations {busy / 1e6:.2f} ms  idle between kernels {idle / 1e6:.2f} ms '
      f'({100.0 * idle / span:.1f} %)  mean gap {idle / max(1, len(gaps)) / 1e3:.2f} us over {len(gaps)} gaps')
gaps.sort()
for q in (0.5, 0.9, 0.99):
    print(f'  gap p{int(q * 100)} {gaps[int(q * (len(gaps) - 1))] / 1e3:.2f} us')
for lo in (20e3, 100e3, 1e6):
    big = [g for g in gaps if g >= lo]
    print(f'  gaps >= {lo / 1e3:.0f} us: {len(big)}  total {sum(big) / 1e6:.3f} ms')
print('largest idle totals, by the kernel BEFORE the gap:')
for k, (c, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f'  {k:42s} gaps {c:5d}  total {t / 1e6:7.3f} ms  mean {t / c / 1e3:6.2f} us')
print('the 25 largest gaps (us): kernel before -> kernel after')
big = []
cur_end = rows[0][1]
for i in range(1, len(rows)):
    g = rows[i][0] - cur_end
    if g > 0:
        big.append((g, rows[i - 1][2].split('(')[0][-44:], rows[i][2].split('(')[0][-44:], (rows[i][0] - rows[0][0]) / 1e6))
    cur_end = max(cur_end, rows[i][1])
for g, a, b, t in sorted(big, reverse=True)[:25]:
    print(f'  {g / 1e3:9.1f}  at {t:8.2f} ms  {a:44s} -> {b}')
