#!/bin/bash
# generic (run-time shape) instantiations of k_mega: FFN loop forms, same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r04f; mkdir -p $OUT
for w in "ecg187 512" "nasa 512" "mimic24 2048"; do for v in gr3 base; do
  if [ $v = base ]; then unset FDIFF_LIB; else export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_$v.so; fi
  echo "$v: $(python scripts/shape_bench.py sample $w 100 2>&1 | tail -1)"; done; done 2>&1 | tee $OUT/generic.txt
