#!/bin/bash
# training-path experiments: one line per (env, lib) combination + kernel stats.  usage: bash scripts/gpu_train_exp.sh TAG
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 nasdaq ms/step', round(d['ms_per_step'],3), d['roofline']['kernel'][:12], round(d['roofline']['avg_kernel_us'],1), 'us')"; }
stats() {  # name, env...
  n=$1; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/bench.py --mode train --no-cpu-baseline > $OUT/$n.log 2>&1)
  echo "== $n: $@"; python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/$n/s_kernel_stats.csv ${NK:-7} | cut -c1-70,100-140
}
statse() {  # name, env...
  n=$1; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train ecg 64 > $OUT/$n.log 2>&1)
  echo "== $n: $@"; python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/$n/s_kernel_stats.csv ${NK:-7} | cut -c1-70,100-140
}
python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_train.py tests/test_gpu_backbones.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for ts in 16 24 32; do
FDIFF_TR_TS=$ts python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 | line "TS=$ts"
FDIFF_TR_TS=$ts python scripts/shape_bench.py train ecg 64 2>/dev/null | tail -1
done
NK=24 stats serial FDIFF_TR_SERIAL=1
stats serial32 FDIFF_TR_SERIAL=1 FDIFF_TR_TS=32
statse ecg_serial32 FDIFF_TR_SERIAL=1 FDIFF_TR_TS=32
