#!/bin/bash
# the reference's droughts shape (T = 365, C = 13): per-layer path (T > 256), sampling and training; kernel stats of both
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_droughts; mkdir -p $OUT
for B in 512 2048; do timeout 300 python scripts/shape_bench.py sample droughts $B 100 2>&1 | tail -1; done
timeout 300 python scripts/shape_bench.py train droughts 64 2>&1 | tail -1
FDIFF_TRAIN_PRECISION=f32 timeout 300 python scripts/shape_bench.py train droughts 64 2>&1 | tail -1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sample -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py sample droughts 512 100 > $OUT/sample.log 2>&1)
python scripts/kstats.py $OUT/sample/s_kernel_stats.csv 8 | cut -c1-70,100-140
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train droughts 64 > $OUT/train.log 2>&1)
python scripts/kstats.py $OUT/train/s_kernel_stats.csv 10 | cut -c1-70,100-140
