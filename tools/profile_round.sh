set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o s -- python3 bench.py --no-alt-math > gpurun_out/b_under_rocprof.json 2> gpurun_out/b_under_rocprof.err
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --no-kernel-timing > gpurun_out/pmc_f.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --no-kernel-timing > gpurun_out/pmc_w.log 2>&1
echo write done
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_summary.json > /dev/null
find gpurun_out/prof_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats.csv \;
rm -rf gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write
ls -la gpurun_out | tail -5
