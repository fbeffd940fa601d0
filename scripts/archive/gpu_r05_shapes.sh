#!/bin/bash
# The reference's dataset shapes on the persistent kernel: run-time-shape instantiation of the library (FDIFF_MEGA_JIT=0) against the
# hiprtc ShapeStatic instantiation (auto policy: 200-step sampler runs), alternating on ONE box.  usage: bash scripts/archive/gpu_r05_shapes.sh [TAG]
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export FDIFF_CACHE_DIR=/tmp/fdiff_jit_cache FDIFF_MEGA_JIT_VERBOSE=1
F=$OUT/mega_dataset_shapes_ab.txt
: > $F
for spec in "ecg187 512" "nasdaq5 512" "mimic24 2048" "nasa251 512" "nasa134 512" "ecg 200" "mimic24 512"; do
  set -- $spec
  for rep in 1 2; do
    for jit in 0 auto; do
      if [ $jit = auto ]; then unset FDIFF_MEGA_JIT; else export FDIFF_MEGA_JIT=$jit; fi
      timeout 300 python scripts/shape_bench.py sample $1 $2 200 2>> $OUT/shapes.err | tail -1 >> $F
    done
  done
done
cat $F
grep "compiled in" $OUT/shapes.err | sed 's/.*k_mega/k_mega/' > $OUT/mega_jit_compile_times.txt; cat $OUT/mega_jit_compile_times.txt
