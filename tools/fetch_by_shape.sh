# usage (GPU box): bash tools/fetch_by_shape.sh [M list]  -> FETCH_SIZE x 2 per launch of the K sweep's shapes (8 x 128^2 px, 1x1, K = 64 ... 2048)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/shape_fetch -o f -- python3 tools/gemm_k_sweep.py --m ${1:-512,2048} --reps 5 > gpurun_out/shape.log 2>&1
python3 - ${1:-512,2048} <<'PY'
import csv, glob, sys
ms = [int(v) for v in sys.argv[1].split(',')]
by = {}
for p in glob.glob('gpurun_out/shape_fetch/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'conv_igemm_f16x3' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            by[int(r['Dispatch_Id'])] = by.get(int(r['Dispatch_Id']), 0.0) + float(r['Counter_Value'])
vals = [v for _, v in sorted(by.items())]
i = 0
for m in ms:
    for k in (64, 128, 256, 512, 1024, 2048):
        chunk = vals[i:i + 8]; i += 8
        alg = (8 * k * 16384 * 4 + m * k * 4) / 1e6
        print('M=%5d K=%5d  FETCH x2 per launch %8.1f MB   algorithmic reads %7.1f MB  (%.2fx)' % (m, k, 2 * sum(chunk[3:]) / 5 * 1024 / 1e6, alg, 2 * sum(chunk[3:]) / 5 * 1024 / 1e6 / alg))
PY
rm -rf gpurun_out/shape_fetch
