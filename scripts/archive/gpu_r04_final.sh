#!/bin/bash
# end-of-round evidence: full GPU suite, the round script (bench line + kernel stats + training), and rocprofv3 kernel stats of the
# 1000-step configs[4] run (the profile the bench row's per-kernel numbers are to be held against).  usage: bash scripts/archive/gpu_r04_final.sh TAG
TAG=${1:-r04d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout 2000 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/gpu_suite.txt; cat $OUT/gpu_suite.txt
bash scripts/gpu_round.sh $TAG 2>&1 | cut -c1-300 | tail -60
cd $GRAFT_REPO_ROOT
python bench.py --workload long --no-cpu-baseline --no-secondary > $OUT/bench_long.json 2> $OUT/bench_long.err; tail -1 $OUT/bench_long.json | cut -c1-500
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_long -o long -- python $GRAFT_REPO_ROOT/bench.py --workload long --no-cpu-baseline --no-secondary > $OUT/stats_long.log 2>&1)
python scripts/kstats.py $OUT/stats_long/long_kernel_stats.csv 5
rm -f $OUT/stats_long/long_kernel_trace.csv
