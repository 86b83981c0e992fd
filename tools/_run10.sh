python -m pytest tests/test_layer_backward_gpu.py tests/test_hip_ops.py tests/test_train_step_gpu.py::test_two_train_steps_match_reference_golden_and_oracle tests/test_fullsize_gpu.py::test_forward_at_baseline_tile_size_matches_oracle tests/test_eval_gpu.py -q -x > gpurun_out/r5_t6.log 2>&1; echo rc=$? >> gpurun_out/r5_t6.log; tail -4 gpurun_out/r5_t6.log
one() { python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt-math 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; h=d['hbm_kernels']
print('$1', round(d['ms_per_step'],2), round(d['alt_single_stream']['ms_per_step'],2), 'gemm', k.get('conv_igemm_f16x3_kernel'), 'wgrad', k.get('conv_wgrad_f16x3_kernel'), 'bn_apply', h['pfst_bn_apply']['ms_per_step'], 'bn_bwd', h['pfst_bn_backward']['ms_per_step'], 'wino_in', h['pfst_wino_input']['ms_per_step'], 'wino_out', h['pfst_wino_output']['ms_per_step'])"; }
for rep in 1 2; do
for L in new bwdnt resnt; do
  if [ $L = new ]; then unset PFST_HIP_LIB; else export PFST_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libpfst_hip_$L.so; fi
  one $L
done; done > gpurun_out/r5_ab_cache_policy2.txt 2>&1
cat gpurun_out/r5_ab_cache_policy2.txt
