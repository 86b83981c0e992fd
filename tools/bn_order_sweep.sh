for O in 0 1 2 4 6 3 7 0; do PFST_BN_ORDER=$O python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt-math 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('order $O', round(d['value'],3), round(d['ms_per_step'],2), 'bn_apply', k.get('pfst_bn_apply'), 'bn_backward', k.get('pfst_bn_backward'))"; done
