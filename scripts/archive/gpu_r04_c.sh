#!/bin/bash
# round 4, call C: pair-form FFN (32x32x16 H) in the static-shape instantiations of k_mega: parity of every static shape, then
# same-box A/B against the 16x16x32 form (libfdiff_hip_f16.so = -DFD_MEGA_FFN32=0)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r04c; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_bench_instantiation.py tests/test_gpu_baseline_shapes.py tests/test_gpu_sampler_parity_shapes.py -m gpu -x -q 2>&1 | tail -5
bash scripts/gpu_variants.sh f16 base f16 base 2>&1 | tee $OUT/variants.txt | grep -E "==|step_ms|FFN loop"
for w in mimic nasdaq; do for v in f16 base; do
  if [ $v = base ]; then unset FDIFF_LIB; else export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_$v.so; fi
  echo "$w $v: $(python scripts/shape_bench.py sample $w 512 100 2>&1 | tail -1)"; done; done 2>&1 | tee $OUT/shapes.txt
