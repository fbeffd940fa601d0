#!/bin/bash
# phase clocks only (variant libraries built with -DFD_TRP_PROF): usage: bash scripts/gpu_r06_clk.sh TAG "variants" "shapes"
TAG=${1:-r06}; VARS=${2:-trpprof}; SHAPES=${3:-"nasdaq ecg"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
for v in $VARS; do for shp in $SHAPES; do
  echo "== $v $shp" | tee -a $OUT/phase_clocks.txt; FDIFF_BENCH_NREP=40 FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_$v.so python scripts/shape_bench.py train $shp 64 2>&1 | grep -v amdgpu.ids | tee -a $OUT/phase_clocks.txt | cut -c1-700
done; done
