# usage (GPU box): bash tools/fetch_calibration.sh   -> gpurun_out/fetch_calibration.txt
# Calibrates rocprofv3's FETCH_SIZE for the f16x3 GEMM's activation loads (coalesced DWORD buffer loads; MI355X_MICROARCH.md calibrates only
# 16-byte-per-lane loads: "double it"; other widths: "calibrate on a known byte count in your own access pattern"): a 1x1 convolution
# 2048 -> 128 over 8 x 128^2 pixels reads its 1.074 GB of activations exactly once (one row tile, no reuse, tensor >> Infinity Cache).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/cal_fetch -o f -- python3 tools/gemm_k_sweep.py --m 128 --reps 5 > gpurun_out/cal.log 2>&1
python3 - <<'PY' > gpurun_out/fetch_calibration.txt
import csv, glob
rows = []
for p in glob.glob('gpurun_out/cal_fetch/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(p)))
by = {}
for r in rows:
    if 'conv_igemm_f16x3' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
        by.setdefault(int(r['Dispatch_Id']), 0.0)
        by[int(r['Dispatch_Id'])] += float(r['Counter_Value'])
vals = [v for _, v in sorted(by.items())]
# the sweep runs K = 64, 128, 256, 512, 1024, 2048 with 3 + 5 launches each: the last 8 dispatches are K = 2048
print('FETCH_SIZE (KiB) of the last 8 dispatches (K = 2048, M = 128, 8 x 128^2 px; activations 1,073,741,824 B + weights 1 MB):')
for v in vals[-8:]:
    print('  raw %.1f MB   x2 = %.1f MB   (expected 1074.8 MB)' % (v * 1024 / 1e6, 2 * v * 1024 / 1e6))
PY
rm -rf gpurun_out/cal_fetch
cat gpurun_out/fetch_calibration.txt
