#!/bin/bash
# phase clocks of k_tr_fwd_layers (variant libraries built with -DFD_TRP_PROF) + step times of the shipped library
# usage: bash scripts/gpu_r06_prof.sh TAG "variant names" [shapes]
TAG=${1:-r06b}; VARS=${2:-trpprof}; SHAPES=${3:-"nasdaq ecg"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
for v in $VARS; do for shp in $SHAPES; do
  echo "== $v $shp" | tee -a $OUT/phase_clocks.txt; FDIFF_BENCH_NREP=40 FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_$v.so python scripts/shape_bench.py train $shp 64 2>&1 | grep -v amdgpu.ids | tee -a $OUT/phase_clocks.txt | cut -c1-400
done; done
timeout 600 python -m pytest tests/test_gpu_train_persist.py -m gpu -x -q 2>&1 | tail -3 | cut -c1-300
for shp in $SHAPES; do
bash scripts/gpu_ab.sh $TAG 2 "FDIFF_TR_PERSIST=0" "FDIFF_TR_PERSIST=1" -- python scripts/shape_bench.py train $shp 64
done
