# usage (GPU box): bash tools/ab_lib.sh <variant> "<command>"      e.g.  bash tools/ab_lib.sh nostore "python tools/gemm_k_sweep.py --m 512"
# Same-box A/B of the in-tree library against ab_libs/libpfst_hip_<variant>.so (python -m pfst_amd.build --variant <variant> <flags>):
# runs <command> as new / variant / new / variant.
V="$1"; shift
for L in new $V new $V; do
  if [ $L = new ]; then unset PFST_HIP_LIB; else export PFST_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libpfst_hip_$V.so; fi
  echo "== $L"
  eval "$@" 2>/dev/null
done
