#!/bin/bash
# configs[4] attention staging: previous commit's library (one tile of rows ahead) / working tree (ring of 4 tiles) / working tree + priority
# for the second wave of every SIMD, alternating on one box (100-step sampler runs), then the phase clocks.  usage: ... [TAG]
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/fourierdiffusion_amd
F=$OUT/attn_long_prefetch_ab.txt
: > $F
for rep in 1 2 3; do
  for v in prev base attnprio; do
    if [ $v = base ]; then unset FDIFF_LIB; else export FDIFF_LIB=$L/libfdiff_hip_$v.so; fi
    echo "$v: $(python scripts/shape_bench.py sample long 64 100 2>&1 | tail -1 | cut -c1-120)" | tee -a $F
  done
done
FDIFF_LIB=$L/libfdiff_hip_attnprof.so python scripts/shape_bench.py sample long 64 5 2>&1 | grep -E "attn dbg" | cut -c1-200 | tee $OUT/attn_long_phase_clocks.txt
