# usage on the GPU box: bash tools/f16x3_counters.sh   -> gpurun_out/f16x3_pmc_*.txt : SQ counter passes over tools/f16x3_probe.py --bench-only
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for CTR in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d gpurun_out/f16pmc_$i -o p -- python3 tools/f16x3_probe.py --bench-only > gpurun_out/f16pmc_$i.log 2>&1 || echo "pass $i failed"
  python3 tools/pmc_fold.py gpurun_out/f16pmc_$i > gpurun_out/f16x3_pmc_$i.txt || true
  rm -rf gpurun_out/f16pmc_$i
done
