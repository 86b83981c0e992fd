# usage (GPU box): bash tools/ab_bench.sh <variant> [steps]     same-box A/B of the whole step: in-tree library vs ab_libs/libpfst_hip_<variant>.so,
# alternating (new / variant / new / variant), `value` only (no CPU baseline, no kernel timing, no other arithmetics)
V="$1"; S="${2:-8}"
for L in new $V new $V; do
  if [ $L = new ]; then unset PFST_HIP_LIB; else export PFST_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libpfst_hip_$V.so; fi
  python bench.py --steps $S --warmup 3 --no-cpu-baseline --no-kernel-timing --no-alt-math 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'])"
done
