cd $GRAFT_REPO_ROOT
python scripts/shape_bench.py train ecg 64 2>&1 | tail -1
python scripts/shape_bench.py train nasdaq 64 2>&1 | tail -1
FDIFF_TRAIN_PRECISION=fp32 python scripts/shape_bench.py train ecg 64 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -o tr -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train ecg 64 > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -30 {}'
