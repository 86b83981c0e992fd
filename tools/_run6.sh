python -m pytest tests/test_hip_ops.py -q -x -k "f16x3 or winograd or chain or minmax" > gpurun_out/r5_t4.log 2>&1; echo rc=$? >> gpurun_out/r5_t4.log; tail -4 gpurun_out/r5_t4.log
for v in 0 1 0 1; do echo "== DEFER_STORE=$v"; PFST_F16X3_DEFER_STORE=$v python tools/gemm_k_sweep.py --m 256,1024,2048 --reps 20 2>/dev/null; done > gpurun_out/r5_defer_sweep.txt
bash tools/ab_env.sh "PFST_F16X3_DEFER_STORE=0" > gpurun_out/r5_ab_defer_store.txt 2>&1; cat gpurun_out/r5_ab_defer_store.txt
