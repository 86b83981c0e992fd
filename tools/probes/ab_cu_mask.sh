# usage (GPU box): bash tools/probes/ab_cu_mask.sh        the weight-gradient side stream confined to a subset of the compute units
# (PFST_WGRAD_CU_MASK, hipExtStreamCreateWithCUMask) against the default high-priority stream; one bench.py run (8 steps) per line
run() {
  L="$1"; M="$2"
  if [ -n "$M" ]; then export PFST_WGRAD_CU_MASK=$M; else unset PFST_WGRAD_CU_MASK; fi
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-alt-math 2>gpurun_out/cu_mask_$L.err | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value'], 3), round(d['ms_per_step'], 2))" || echo "$L failed: $(tail -2 gpurun_out/cu_mask_$L.err)"
}
H=0000ffff,0000ffff,0000ffff,0000ffff,0000ffff,0000ffff,0000ffff,0000ffff
Q3=00ffffff,00ffffff,00ffffff,00ffffff,00ffffff,00ffffff,00ffffff,00ffffff
Q1=000000ff,000000ff,000000ff,000000ff,000000ff,000000ff,000000ff,000000ff
LO=ffffffff,ffffffff,ffffffff,ffffffff,0,0,0,0
EV=55555555,55555555,55555555,55555555,55555555,55555555,55555555,55555555
ALL=ffffffff,ffffffff,ffffffff,ffffffff,ffffffff,ffffffff,ffffffff,ffffffff
run default ""
run half16 $H
run threeq $Q3
run quarter $Q1
run default ""
run lowhalf $LO
run even $EV
run all $ALL
run half16 $H
