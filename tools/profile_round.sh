# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r02
# Produces under gpurun_out/: <R>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the default bench command),
# <R>_bench_under_rocprof.json, <R>_pmc_hbm_traffic_per_launch.json (FETCH_SIZE / WRITE_SIZE, separate passes),
# <R>_mfma_busy.json (SQ instruction / busy counters per kernel).  Copy what is to be judged into profiles/.
# The program sits directly after `--` (no env / bash -c hop: the profiler's preloaded library has initialised the GPU).
set -e
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# Per-kernel durations are only comparable with bench.py's `roofline` (measured in its single-stream region) when nothing co-runs: the stats and
# counter passes run the whole command on ONE stream (exported, not `env`: the program must sit directly after `--`); a second stats pass
# records the product schedule (teacher pass and weight gradients on side streams) for the co-running picture.
export PFST_WGRAD_STREAM=0 PFST_FORK_TEACHER=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o s -- python3 bench.py --no-alt-math > gpurun_out/${R}_bench_under_rocprof.json 2> gpurun_out/b_under_rocprof.err
echo stats done
unset PFST_WGRAD_STREAM PFST_FORK_TEACHER
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats_ov -o s -- python3 bench.py --no-alt-math --no-cpu-baseline > gpurun_out/${R}_bench_under_rocprof_overlap.json 2> gpurun_out/b_under_rocprof_ov.err
find gpurun_out/prof_stats_ov -name "*kernel_stats.csv" -exec cp {} gpurun_out/${R}_kernel_stats_overlap.csv \;
rm -rf gpurun_out/prof_stats_ov
echo overlap stats done
export PFST_WGRAD_STREAM=0 PFST_FORK_TEACHER=0
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --no-kernel-timing"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py $ARGS > gpurun_out/pmc_f.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py $ARGS > gpurun_out/pmc_w.log 2>&1
echo write done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq -o q -- python3 bench.py $ARGS > gpurun_out/pmc_q.log 2>&1
echo sq done
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/${R}_pmc_hbm_traffic_per_launch_${MATH:-f16x3}.json > /dev/null
python3 tools/pmc_mfma.py gpurun_out/pmc_sq gpurun_out/${R}_mfma_busy.json > /dev/null
find gpurun_out/prof_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${R}_kernel_stats.csv \;
rm -rf gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
ls -la gpurun_out | tail -8
