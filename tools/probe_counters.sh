# usage on the GPU box: bash tools/probe_counters.sh <tag> "<counters>" <kernel_probe args...>   -> gpurun_out/probe_<tag>.txt
# One rocprofv3 counter pass over tools/kernel_probe.py (a few launches of chosen conv kernels at BASELINE shapes).
set -e
TAG=$1; shift
CTR=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d gpurun_out/probe_$TAG -o p -- python3 tools/kernel_probe.py "$@" > gpurun_out/probe_$TAG.log 2>&1
python3 tools/pmc_fold.py gpurun_out/probe_$TAG > gpurun_out/probe_$TAG.txt
rm -rf gpurun_out/probe_$TAG
