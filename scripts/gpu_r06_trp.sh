#!/bin/bash
# Round 6, persistent training forward: parity tests first, then the step time against the per-layer kernels on the same box.
TAG=${1:-r06a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
rm -f $GRAFT_REPO_ROOT/gpurun_out/parity_errors.log
timeout 900 python -m pytest tests/test_gpu_train_persist.py -m gpu -x -q 2>&1 | tail -25 | cut -c1-400 | tee $OUT/persist_tests.txt
cp $GRAFT_REPO_ROOT/gpurun_out/parity_errors.log $OUT/parity_errors.txt 2>/dev/null
bash scripts/gpu_ab.sh $TAG 2 "FDIFF_TR_PERSIST=0" "FDIFF_TR_PERSIST=1" "FDIFF_TR_PERSIST=2" -- python scripts/shape_bench.py train nasdaq 64
bash scripts/gpu_ab.sh $TAG 2 "FDIFF_TR_PERSIST=0" "FDIFF_TR_PERSIST=1" "FDIFF_TR_PERSIST=1 FDIFF_TR_PERSIST_NT=4" -- python scripts/shape_bench.py train ecg 64
