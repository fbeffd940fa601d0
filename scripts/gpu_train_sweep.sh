#!/bin/bash
# Training-path loop on the GPU box: tests, token-split sweep of k_tr_wgrad, kernel stats.  usage: bash scripts/gpu_train_sweep.sh TAG "4 8 12 16"
TAG=${1:-t}
SWEEP=${2:-"16"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -5
for ts in $SWEEP; do
  for mode in nasdaq ecg; do
    if [ $mode = nasdaq ]; then
      FDIFF_TR_TS=$ts python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TS=$ts nasdaq ms/step', round(d['ms_per_step'],3), d['roofline']['kernel'][:12], round(d['roofline']['avg_kernel_us'],1), 'us frac', round(d['roofline']['frac'],4))"
    else
      FDIFF_TR_TS=$ts python scripts/shape_bench.py train ecg 64 2>/dev/null | tail -1
    fi
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o train_nasdaq -- python $GRAFT_REPO_ROOT/bench.py --mode train --no-cpu-baseline > $OUT/stats.log 2>&1
python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/stats/train_nasdaq_kernel_stats.csv 10
