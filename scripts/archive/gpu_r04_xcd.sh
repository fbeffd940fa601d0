#!/bin/bash
# round 4: XCD-aware re-dealing of workgroup ids in the training kernels (FDIFF_TR_XCD=1, default) against the hardware order (=0).
TAG=${1:-xcd}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
sb() { python scripts/shape_bench.py train $1 64 2>/dev/null | tail -1 | cut -c1-110; }
stats() {  # name, shape, env...
  n=$1; shp=$2; shift; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train $shp 64 > $OUT/$n.log 2>&1)
  echo "== $n: $@"; python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/$n/s_kernel_stats.csv ${NK:-7} | cut -c1-70,100-140
}
timeout 900 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_benched_shapes.py tests/test_gpu_widths.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for rep in 1 2 3; do
echo "xcd=1 nasdaq: $(sb nasdaq)"
echo "xcd=0 nasdaq: $(FDIFF_TR_XCD=0 sb nasdaq)"
echo "xcd=1 ecg:    $(sb ecg)"
echo "xcd=0 ecg:    $(FDIFF_TR_XCD=0 sb ecg)"
done
stats serial_xcd1 nasdaq FDIFF_TR_SERIAL=1
stats serial_xcd0 nasdaq FDIFF_TR_SERIAL=1 FDIFF_TR_XCD=0
stats ecg_serial_xcd1 ecg FDIFF_TR_SERIAL=1
stats ecg_serial_xcd0 ecg FDIFF_TR_SERIAL=1 FDIFF_TR_XCD=0
