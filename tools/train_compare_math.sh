# usage (GPU box): bash tools/train_compare_math.sh r03 -> gpurun_out/<R>_train_compare_math.txt
# The same seeded synthetic 120-iteration training run (random init, b=2, 256x256 crops) under the three arithmetics of the dense
# convolutions -- fp32-input MFMA, bf16x6, f16x3 (the default) --, losses side by side every 10 iterations.
set -e
R=${1:-r03}
ARGS="pfst_pots_irrg2vaih_irrg_deeplabv3plus_r50-d8 --synthetic --random-init --max-iters 120 --batch-size 2 --crop-size 256 --no-validate --seed 0 --cfg-options optimizer.lr=0.0005 lr_config.warmup_iters=20 log_config.interval=10"
PFST_CONV_MATH=f32 python3 tools/train.py $ARGS --work-dir /tmp/w_f32 > /tmp/train_f32.log 2>&1
PFST_CONV_MATH=bf16x6 python3 tools/train.py $ARGS --work-dir /tmp/w_b6 > /tmp/train_b6.log 2>&1
PFST_CONV_MATH=f16x3 python3 tools/train.py $ARGS --work-dir /tmp/w_f16 > /tmp/train_f16.log 2>&1
python3 - <<PY > gpurun_out/${R}_train_compare_math.txt
import re
def rows(p):
    out = {}
    for l in open(p):
        m = re.match(r'.*Iter \[(\d+)/', l)
        if m:
            out[int(m.group(1))] = {k: float(v) for k, v in re.findall(r'([a-z_.]+): (-?[\d.]+(?:e-?\d+)?)', l)}
    return out
a, b, c = rows('/tmp/train_f32.log'), rows('/tmp/train_b6.log'), rows('/tmp/train_f16.log')
keys = ['decode.loss_ce', 'decode.acc_seg', 'mix.decode.loss_ce', 'loss_src_pos_mean', 'loss_sim_pos']
print('same seeded synthetic run: fp32-input MFMA | bf16x6 | f16x3 (PFST_CONV_MATH)')
print('iter  ' + '  '.join('%-38s' % k for k in keys))
for it in sorted(a):
    if it in b and it in c:
        print('%4d  ' % it + '  '.join('%11.4f |%11.4f |%11.4f ' % (a[it].get(k, float('nan')), b[it].get(k, float('nan')), c[it].get(k, float('nan'))) for k in keys))
PY
# --deterministic (round 5): the same f16x3 run twice with every floating-point summation order fixed -- the log lines must be identical
PFST_CONV_MATH=f16x3 python3 tools/train.py $ARGS --deterministic --work-dir /tmp/w_d1 > /tmp/train_d1.log 2>&1
PFST_CONV_MATH=f16x3 python3 tools/train.py $ARGS --deterministic --work-dir /tmp/w_d2 > /tmp/train_d2.log 2>&1
python3 - <<PY >> gpurun_out/${R}_train_compare_math.txt
import re
def rows(p):
    return [re.sub(r'^.*?(Iter \\[)', r'\\1', re.sub(r'(time|data_time|memory|eta): [^,]*,? ?', '', l)).strip() for l in open(p) if re.match(r'.*Iter \\[\\d+/', l)]
a, b = rows('/tmp/train_d1.log'), rows('/tmp/train_d2.log')
print()
print('--deterministic, two runs of the f16x3 training: %d log lines each, %s' % (len(a), 'IDENTICAL in every logged value' if a == b and a else 'DIFFERENT'))
print('last line: ' + (a[-1] if a else '-'))
PY
cat gpurun_out/${R}_train_compare_math.txt
