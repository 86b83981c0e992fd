for L in new nt sc1 same new nt sc1 same; do
  if [ $L = new ]; then unset PFST_HIP_LIB; else export PFST_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libpfst_hip_$L.so; fi
  echo "== $L"
  python tools/gemm_k_sweep.py --m 1024,2048 --reps 20 2>/dev/null
done > gpurun_out/r5_store_policy_sweep.txt
unset PFST_HIP_LIB
bash tools/ab_lib.sh nt "python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-alt-math | python -c \"import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['alt_single_stream']['ms_per_step'],2), d['kernel_ms_per_step']['conv_igemm_f16x3_kernel'])\"" > gpurun_out/r5_ab_nt.txt 2>&1
cat gpurun_out/r5_ab_nt.txt
