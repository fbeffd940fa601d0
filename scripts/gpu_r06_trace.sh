#!/bin/bash
# rocprofv3 kernel trace + stats of the training step (persistent forward on / off): usage: bash scripts/gpu_r06_trace.sh TAG [shape]
TAG=${1:-r06t}; SHP=${2:-nasdaq}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  FDIFF_TR_PERSIST=$mode FDIFF_BENCH_NREP=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$mode -o t -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train $SHP 64 > $OUT/p$mode.log 2>&1
  tail -1 $OUT/p$mode.log | cut -c1-200
  python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/p$mode/t_kernel_stats.csv 14 | cut -c1-150
done
python $GRAFT_REPO_ROOT/scripts/step_timeline.py $OUT/p1/t_kernel_trace.csv | tail -80
rm -f $OUT/p0/t_kernel_trace.csv
