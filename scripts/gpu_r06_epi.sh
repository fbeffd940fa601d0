#!/bin/bash
# k_tr_ffn_bwd epilogue split between the two waves of a tile: training tests + step time + phase clocks when the variant library exists
TAG=${1:-r06epi}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_train.py tests/test_gpu_train_persist.py tests/test_gpu_benched_shapes.py -x -q -m gpu > $OUT/tests.log 2>&1
grep -E "passed|failed" $OUT/tests.log | tail -2
grep -E "^E  |Error" $OUT/tests.log | head -8 | cut -c1-300
if [ -f fourierdiffusion_amd/libfdiff_hip_fbprof.so ]; then bash scripts/gpu_r06_clk.sh $TAG fbprof "nasdaq ecg" | grep -A3 "phase clocks" | cut -c1-420; fi
for shp in nasdaq ecg; do
  bash scripts/gpu_ab.sh $TAG 3 "A=1" -- python scripts/shape_bench.py train $shp 64 | cut -c1-150
done
