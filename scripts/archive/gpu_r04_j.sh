#!/bin/bash
# fused unembed + SDE step + next embedding for T > 256: parity, then the configs[4] row with and without it (same box)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r04j; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_benched_shapes.py tests/test_gpu_baseline_shapes.py -m gpu -x -q 2>&1 | tail -5
for i in 1 2; do
  echo "fused:   $(python scripts/shape_bench.py sample long 64 100 2>&1 | tail -1)"
  echo "nopair:  $(FDIFF_FFN_NO_PAIR=1 python scripts/shape_bench.py sample long 64 100 2>&1 | tail -1)"
done 2>&1 | tee $OUT/long_ab.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -o long -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py sample long 64 100 > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1
python $GRAFT_REPO_ROOT/scripts/kstats.py $GRAFT_REPO_ROOT/$OUT/stats/long_kernel_stats.csv 8
