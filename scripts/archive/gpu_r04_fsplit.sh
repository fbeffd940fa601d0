#!/bin/bash
# F-split of the training FFN kernels (pairs of workgroups per 64-token block when 2 x blocks <= CUs): parity, T=100 A/B, solo kernel times
OUT=$GRAFT_REPO_ROOT/gpurun_out/fsplit
mkdir -p $OUT
sb() { python scripts/shape_bench.py train $1 $2 2>/dev/null | tail -1 | cut -c1-110; }
timeout 900 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_benched_shapes.py tests/test_gpu_widths.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for rep in 1 2 3; do
echo "split    ecg B=64: $(sb ecg 64)"
echo "no split ecg B=64: $(FDIFF_TR_FSPLIT=0 sb ecg 64)"
done
echo "split    ecg B=32: $(sb ecg 32)"
echo "no split ecg B=32: $(FDIFF_TR_FSPLIT=0 sb ecg 32)"
echo "split    nasdaq B=16: $(sb nasdaq 16)"
echo "no split nasdaq B=16: $(FDIFF_TR_FSPLIT=0 sb nasdaq 16)"
for v in 1 0; do
  n=ecg_serial_fsplit$v
  (cd /tmp && export TMPDIR=/tmp && FDIFF_TR_SERIAL=1 FDIFF_TR_FSPLIT=$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train ecg 64 > $OUT/$n.log 2>&1)
  echo "== $n"; python3 $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/$n/s_kernel_stats.csv 6 | cut -c1-60,100-140
done
