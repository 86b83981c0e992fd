# usage (GPU box): bash tools/ab_env.sh "PFST_WGRAD_STREAM=0 PFST_FORK_TEACHER=0"
# (round 5 removed the per-fold PFST_* switches once their A/B was on file: profiles/r04_ab_*.txt, r05_ab_*.txt; what is left to flip at run time
# is listed in README.md -- a fold that needs a new A/B gets a -D variant build instead: python -m pfst_amd.build --variant NAME -DFLAG, tools/ab_lib.sh)
# Same-box A/B of run-time switches of ONE build (boxes of the pool differ by 2-3 %): alternates the given environment (`off`) with the
# default (`on`) over four bench.py runs and prints the step plus the kernels the switches touch.
OFF="$1"
for L in off on off on; do
  if [ $L = off ]; then PRE="env $OFF"; else PRE=""; fi
  $PRE python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt-math 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; h=d['hbm_kernels']
print('$L', round(d['value'],3), round(d['ms_per_step'],2), {n:h[n]['ms_per_step'] for n in ('pfst_bn_apply','pfst_bn_backward')}, {n:v for n,v in k.items() if 'wgrad' in n})"
done
