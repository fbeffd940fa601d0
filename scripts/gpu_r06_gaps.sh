#!/bin/bash
# kernel trace of one optimizer step with names, queue ids and the idle gaps of the main queue: bash scripts/gpu_r06_gaps.sh TAG [shape] [env...]
TAG=${1:-r06g}; SHP=${2:-nasdaq}; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" FDIFF_BENCH_NREP=20 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train $SHP 64 > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-160
python $GRAFT_REPO_ROOT/scripts/step_timeline.py $OUT/t/t_kernel_trace.csv > $OUT/timeline.txt
rm -f $OUT/t/t_kernel_trace.csv
cat $OUT/timeline.txt
