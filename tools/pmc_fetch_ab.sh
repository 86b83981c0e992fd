# usage (GPU box): bash tools/pmc_fetch_ab.sh <kernel-name-substring>   (PFST_HIP_LIB selects the build) -> FETCH_SIZE per launch of matching kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${2:-new}
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ab_fetch_$TAG -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --no-kernel-timing > gpurun_out/ab_$TAG.log 2>&1
python3 - "$1" $TAG <<'PY'
import csv, glob, sys
from collections import defaultdict
sub, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
for p in glob.glob(f'gpurun_out/ab_fetch_{tag}/**/*counter_collection.csv', recursive=True):
    per = defaultdict(float); name = {}
    for r in csv.DictReader(open(p)):
        if r['Counter_Name'] == 'FETCH_SIZE':
            per[r['Dispatch_Id']] += float(r['Counter_Value']); name[r['Dispatch_Id']] = r['Kernel_Name']
    for d, v in per.items():
        k = name[d].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if sub in k:
            acc[k][0] += 1; acc[k][1] += v
for k, (n, v) in sorted(acc.items()):
    print(tag, k[:70], 'launches', n, 'FETCH x2 per launch %.1f MB' % (2 * v * 1024 / n / 1e6))
PY
rm -rf gpurun_out/ab_fetch_$TAG
