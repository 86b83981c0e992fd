# usage (GPU box): bash tools/probes/timeline_overlap.sh     kernel trace of a short run on the PRODUCT schedule -> how much of a step has 0 / 1 / >= 2
# kernels in flight, the largest idle gaps and what surrounds them, and what runs behind the last data-gradient kernel of the step
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -o t -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt-math --no-kernel-timing > gpurun_out/tl_bench.json 2> gpurun_out/tl_bench.err
find gpurun_out/prof_tl -name "*kernel_trace.csv" -exec cp {} gpurun_out/tl_kernel_trace.csv \;
rm -rf gpurun_out/prof_tl
python3 tools/probes/timeline_overlap.py gpurun_out/tl_kernel_trace.csv
rm -f gpurun_out/tl_kernel_trace.csv
