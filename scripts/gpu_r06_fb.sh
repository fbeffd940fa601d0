#!/bin/bash
# k_tr_ffn_bwd prologue: tests of the training path, phase clocks (variant library libfdiff_hip_fbprof.so, -DFD_TR_PROF_FB) and the same-box
# A/B of the row-linear partial sums (FDIFF_TR_FB_ROWSUM): bash scripts/gpu_r06_fb.sh TAG
TAG=${1:-r06fb}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_train.py tests/test_gpu_train_persist.py tests/test_gpu_benched_shapes.py -x -q -m gpu > $OUT/tests.log 2>&1
grep -E "passed|failed" $OUT/tests.log | tail -2
if [ -f fourierdiffusion_amd/libfdiff_hip_fbprof.so ]; then bash scripts/gpu_r06_clk.sh $TAG fbprof "nasdaq ecg" | grep -A3 "phase clocks" | cut -c1-420; fi
for shp in nasdaq ecg; do
  bash scripts/gpu_ab.sh $TAG 3 "FDIFF_TR_FB_ROWSUM=0" "FDIFF_TR_FB_ROWSUM=1" -- python scripts/shape_bench.py train $shp 64 | cut -c1-150
done
