#!/bin/bash
# d x of the attention side as rows . in_proj^T in the consumer (FDIFF_TR_DX_GEMM): training tests + same-box A/B: bash scripts/gpu_r06_dxg.sh TAG
TAG=${1:-r06dxg}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_train.py tests/test_gpu_train_persist.py tests/test_gpu_benched_shapes.py -x -q -m gpu > $OUT/tests.log 2>&1
grep -E "passed|failed" $OUT/tests.log | tail -2
grep -E "^E  |Error" $OUT/tests.log | head -8 | cut -c1-300
for shp in nasdaq ecg; do
  bash scripts/gpu_ab.sh $TAG 3 "FDIFF_TR_DX_GEMM=0" "FDIFF_TR_DX_GEMM=1" -- python scripts/shape_bench.py train $shp 64 | cut -c1-150
done
