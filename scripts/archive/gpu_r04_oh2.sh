#!/bin/bash
# round 4: attention backward forms (FDIFF_TR_ATTN_OH = 0 pair, 1 one head + fp32 parts, 2 one head + bf16 parts), the fixed
# k_tr_masks_T, and one weight-gradient workgroup per CU (FDIFF_TR_WG_LDS_KB=84).  usage: bash scripts/archive/gpu_r04_oh2.sh TAG
TAG=${1:-oh2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 nasdaq ms/step', round(d['ms_per_step'],3), d['roofline']['kernel'][:14], round(d['roofline']['avg_kernel_us'],1), 'us')"; }
stats() {  # name, env...
  n=$1; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/bench.py --mode train --no-cpu-baseline > $OUT/$n.log 2>&1)
  echo "== $n: $@"; python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/$n/s_kernel_stats.csv ${NK:-8} | cut -c1-70,100-140
}
timeout 900 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_benched_shapes.py tests/test_gpu_widths.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
grep -h "parity" gpurun_out/parity_errors.log 2>/dev/null | grep -i "grad" | tail -12
for rep in 1 2; do
for cfg in "FDIFF_TR_ATTN_OH=0" "FDIFF_TR_ATTN_OH=1" "FDIFF_TR_ATTN_OH=2" "FDIFF_TR_ATTN_OH=2 FDIFF_TR_WG_LDS_KB=84" "FDIFF_TR_ATTN_OH=0 FDIFF_TR_WG_LDS_KB=84"; do
env $cfg python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 | line "$cfg"
echo "$cfg ecg: $(env $cfg python scripts/shape_bench.py train ecg 64 2>/dev/null | tail -1 | cut -c40-90)"
done
done
stats serial_oh2 FDIFF_TR_SERIAL=1 FDIFF_TR_ATTN_OH=2
stats par_oh2 FDIFF_TR_ATTN_OH=2
stats par_oh2_wg1 FDIFF_TR_ATTN_OH=2 FDIFF_TR_WG_LDS_KB=84
