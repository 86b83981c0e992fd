# usage (GPU box): bash tools/profile_bf16x6.sh r02   -> gpurun_out/<R>_mfma_busy_bf16x6.json, <R>_per_layer.txt, <R>_per_layer_bf16x6.txt
# SQ counter pass of one step under PFST_CONV_MATH=bf16x6 (the variable is set for the profiler AND the program: no hop after `--`),
# and the per-layer convolution timing table of the default arithmetic.
set -e
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PFST_CONV_MATH=bf16x6
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq6 -o q -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --no-kernel-timing > gpurun_out/pmc_q6.log 2>&1
python3 tools/pmc_mfma.py gpurun_out/pmc_sq6 gpurun_out/${R}_mfma_busy_bf16x6.json > /dev/null
rm -rf gpurun_out/pmc_sq6
python3 bench.py --per-layer --steps 2 --warmup 1 --no-cpu-baseline --no-alt-math > /dev/null 2> gpurun_out/${R}_per_layer_bf16x6.txt
unset PFST_CONV_MATH
python3 bench.py --per-layer --steps 2 --warmup 1 --no-cpu-baseline --no-alt-math > gpurun_out/${R}_per_layer.json 2> gpurun_out/${R}_per_layer.txt
echo done
