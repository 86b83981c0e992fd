# usage (GPU box): bash tools/ab_prev_lib.sh
# Same-box A/B of two builds of the kernel library: boxes of this pool differ by several per cent, so two gpurun calls cannot be compared.
# Build the older commit's library in the container first (git stash / checkout; python -m pfst_amd.build; copy libpfst_hip.so to
# pfst_amd/build_prev/libpfst_hip_prev.so -- *.so is git-ignored but travels with the snapshot), then run this: it alternates
# PFST_HIP_LIB=<prev> and the in-tree build over four bench.py runs and prints the step and the streaming kernels' times.
for L in prev new prev new; do
  if [ $L = prev ]; then export PFST_HIP_LIB=$GRAFT_REPO_ROOT/pfst_amd/build_prev/libpfst_hip_prev.so; else unset PFST_HIP_LIB; fi
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt-math 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['hbm_kernels']
print('$L', round(d['value'],3), round(d['ms_per_step'],2), {k:(h[k]['ms_per_step'], h[k]['frac_of_8TBps']) for k in ('pfst_bn_apply','pfst_bn_backward','pfst_dwconv3x3_wgrad')})"
done
