python -m pytest tests -m gpu -q -x > gpurun_out/r5_t5.log 2>&1; echo rc=$? >> gpurun_out/r5_t5.log; tail -5 gpurun_out/r5_t5.log
for v in 0 1 0 1; do echo "== NT_STORE=$v"; PFST_F16X3_NT_STORE=$v python tools/gemm_k_sweep.py --m 1024,2048 --reps 20 2>/dev/null; done > gpurun_out/r5_nt_sweep.txt
for st in 0 1 2; do echo "== stats=$st"; python tools/gemm_k_sweep.py --m 512,2048 --reps 20 --stats $st 2>/dev/null; done > gpurun_out/r5_stats_sweep.txt
bash tools/ab_env.sh "PFST_F16X3_NT_STORE=0" > gpurun_out/r5_ab_nt_store.txt 2>&1; cat gpurun_out/r5_ab_nt_store.txt
