#!/bin/bash
# A/B of the event packets on the training stream (FDIFF_TR_EXT_EVENT: stop event of k_tr_attn_bwd instead of a record behind it;
# FDIFF_TR_LEAN_EVENTS: bits, the other joins / records of the step; FDIFF_TR_EVENT_FENCE: events with the system-scope fence;
# FDIFF_TR_EAGER_IMAGES: the weight-image rebuild started by the optimizer's wrapper): bash scripts/gpu_r06_ext.sh TAG
TAG=${1:-r06x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_train_persist.py tests/test_gpu_train_bf16.py tests/test_gpu_train.py tests/test_gpu_benched_shapes.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/tests.txt
for shp in nasdaq ecg; do
bash scripts/gpu_ab.sh $TAG 3 "FDIFF_TR_EXT_EVENT=0 FDIFF_TR_LEAN_EVENTS=0 FDIFF_TR_EVENT_FENCE=1 FDIFF_TR_EAGER_IMAGES=0" "FDIFF_TR_EAGER_IMAGES=0 FDIFF_TR_LEAN_EVENTS=0" \
  "FDIFF_TR_EAGER_IMAGES=0" "FDIFF_TR_EAGER_IMAGES=1" "FDIFF_TR_EAGER_IMAGES=1 FDIFF_TR_LEAN_EVENTS=27" "FDIFF_TR_EAGER_IMAGES=1 FDIFF_TR_LEAN_EVENTS=30" -- python scripts/shape_bench.py train $shp 64 | cut -c1-150
done
