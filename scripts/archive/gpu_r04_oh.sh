#!/bin/bash
# round 4: attention backward per (head, series) (FDIFF_TR_ATTN_OH=1, default) against the pair form (=0): parity, step time at
# both training shapes (alternating), solo kernel times.  usage: bash scripts/archive/gpu_r04_oh.sh TAG
TAG=${1:-oh}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 nasdaq ms/step', round(d['ms_per_step'],3), d['roofline']['kernel'][:14], round(d['roofline']['avg_kernel_us'],1), 'us')"; }
stats() {  # name, env...
  n=$1; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/bench.py --mode train --no-cpu-baseline > $OUT/$n.log 2>&1)
  echo "== $n: $@"; python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/$n/s_kernel_stats.csv ${NK:-8} | cut -c1-70,100-140
}
statse() {  # name, env...
  n=$1; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train ecg 64 > $OUT/$n.log 2>&1)
  echo "== $n: $@"; python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/$n/s_kernel_stats.csv ${NK:-8} | cut -c1-70,100-140
}
timeout 900 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_benched_shapes.py tests/test_gpu_widths.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for rep in 1 2 3; do
for oh in 0 1; do
FDIFF_TR_ATTN_OH=$oh python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 | line "OH=$oh"
echo "OH=$oh ecg: $(FDIFF_TR_ATTN_OH=$oh python scripts/shape_bench.py train ecg 64 2>/dev/null | tail -1)"
done
done
stats serial_oh0 FDIFF_TR_SERIAL=1 FDIFF_TR_ATTN_OH=0
stats serial_oh1 FDIFF_TR_SERIAL=1 FDIFF_TR_ATTN_OH=1
stats par_oh1 FDIFF_TR_ATTN_OH=1
statse ecg_serial_oh0 FDIFF_TR_SERIAL=1 FDIFF_TR_ATTN_OH=0
statse ecg_serial_oh1 FDIFF_TR_SERIAL=1 FDIFF_TR_ATTN_OH=1
