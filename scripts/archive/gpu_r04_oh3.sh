#!/bin/bash
# round 4: attention backward with the keep-multiplier table and unmasked loop reads: parity, both training shapes with each form,
# solo kernel times.  usage: bash scripts/archive/gpu_r04_oh3.sh TAG
TAG=${1:-oh3}
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
timeout 900 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_benched_shapes.py tests/test_gpu_widths.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
grep -h "parity" gpurun_out/parity_errors.log 2>/dev/null | grep -i "benched shape\|form" | tail -12
for rep in 1 2; do
for cfg in "FDIFF_TR_ATTN_OH=0" "FDIFF_TR_ATTN_OH=2"; do
env $cfg python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 | line "$cfg"
echo "$cfg ecg: $(env $cfg python scripts/shape_bench.py train ecg 64 2>/dev/null | tail -1 | cut -c40-90)"
done
done
stats serial_oh2 FDIFF_TR_SERIAL=1 FDIFF_TR_ATTN_OH=2
stats serial_oh0 FDIFF_TR_SERIAL=1 FDIFF_TR_ATTN_OH=0
statse ecg_serial_oh0 FDIFF_TR_SERIAL=1 FDIFF_TR_ATTN_OH=0
statse ecg_serial_oh2 FDIFF_TR_SERIAL=1 FDIFF_TR_ATTN_OH=2
