"""step timeline from a rocprofv3 kernel trace (tools/probes/timeline_overlap.sh): device time with 0 / 1 / >= 2 kernels in flight per step, the largest
idle gaps with their neighbours, and the tail behind the last backward kernel"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0],
              r.get('Queue_Id', '?')) for r in rows), key=lambda e: e[0])
# steps end with the optimizer kernel
ends = [i for i, e in enumerate(ev) if e[2].startswith('adamw_kernel')]
print('optimizer launches found:', len(ends), ' kernels:', len(ev))
for si in range(max(1, len(ends) - 2), len(ends)):
    lo, hi = ends[si - 1] + 1, ends[si]
    step = ev[lo:hi + 1]
    t0, t1 = step[0][0], max(e[1] for e in step)
    pts = sorted([(e[0], 1) for e in step] + [(e[1], -1) for e in step])
    depth, last, cover = 0, t0, {0: 0, 1: 0, 2: 0}
    for t, d in pts:
        cover[min(depth, 2)] += t - last
        depth, last = depth + d, t
    span = t1 - t0
    print(f'step {si}: span {span / 1e6:.2f} ms, idle {cover[0] / 1e6:.2f}, one kernel {cover[1] / 1e6:.2f}, two or more {cover[2] / 1e6:.2f}; '
          f'sum of kernel durations {sum(e[1] - e[0] for e in step) / 1e6:.2f} ms, queues {sorted(set(e[3] for e in step))}')
    # idle gaps
    gaps, cur_end, prev = [], step[0][1], step[0]
    for e in step[1:]:
        if e[0] > cur_end:
            gaps.append((e[0] - cur_end, prev[2], e[2]))
        if e[1] > cur_end:
            cur_end, prev = e[1], e
    gaps.sort(reverse=True)
    print('   largest idle gaps (us, kernel before -> after):')
    for g, a, b in gaps[:8]:
        print(f'      {g / 1e3:8.1f}  {a[:50]} -> {b[:50]}')
    print(f'   gaps > 20 us: {sum(1 for g in gaps if g[0] > 20000)}, total of all {len(gaps)} gaps {sum(g[0] for g in gaps) / 1e6:.2f} ms')
    # by queue: busy time
    for q in sorted(set(e[3] for e in step)):
        qs = [e for e in step if e[3] == q]
        print(f'   queue {q}: {len(qs)} kernels, {sum(e[1] - e[0] for e in qs) / 1e6:.2f} ms, first at +{(qs[0][0] - t0) / 1e6:.2f} ms, last ends at +{(max(e[1] for e in qs) - t0) / 1e6:.2f} ms')
    # the tail: what runs after the last igemm (data-gradient / forward) kernel
    last_gemm = max(i for i, e in enumerate(step) if 'conv_igemm' in e[2])
    tail = step[last_gemm + 1:]
    print(f'   behind the last conv_igemm kernel (ends at +{(step[last_gemm][1] - t0) / 1e6:.2f} ms): {len(tail)} kernels, until +{(t1 - t0) / 1e6:.2f} ms:')
    agg = {}
    for e in tail:
        agg[e[2][:48]] = agg.get(e[2][:48], 0) + (e[1] - e[0])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:8]:
        print(f'      {v / 1e3:8.1f} us  {k}')
