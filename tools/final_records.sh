set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_tests.log 2>&1; echo suite rc=$?; tail -2 gpurun_out/r05_gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_20steps.json 2> gpurun_out/r05_bench.err && echo bench done
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1; echo profile rc=$?
PFST_WGRAD_STREAM=0 PFST_FORK_TEACHER=0 python bench.py --steps 4 --warmup 2 --per-layer --no-cpu-baseline --no-alt-math > gpurun_out/r05_per_layer.json 2> gpurun_out/r05_per_layer_f16x3.txt; echo per-layer rc=$?
