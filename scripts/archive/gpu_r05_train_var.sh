#!/bin/bash
# training step: a library variant against the working tree's library, one box, alternating; then solo (FDIFF_TR_SERIAL=1) kernel
# times of both.  usage: bash scripts/archive/gpu_r05_train_var.sh VARIANT [TAG]
V=$1; TAG=${2:-r05_$V}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
VAR=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_$V.so
sb() { timeout 120 python scripts/shape_bench.py train $1 64 2>/dev/null | tail -1 | cut -c1-110; }
stats() {  # name, shape, env...
  n=$1; shp=$2; shift; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train $shp 64 > $OUT/$n.log 2>&1)
  echo "== $n: $@"; python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/$n/s_kernel_stats.csv ${NK:-7} | cut -c1-70,100-140
}
timeout 900 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_benched_shapes.py tests/test_gpu_widths.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for rep in 1 2 3; do
echo "base nasdaq: $(sb nasdaq)"
echo "$V nasdaq: $(FDIFF_LIB=$VAR sb nasdaq)"
echo "base ecg:    $(sb ecg)"
echo "$V ecg:    $(FDIFF_LIB=$VAR sb ecg)"
done
stats serial_base nasdaq FDIFF_TR_SERIAL=1
stats serial_$V nasdaq FDIFF_TR_SERIAL=1 FDIFF_LIB=$VAR
stats serial_base_ecg ecg FDIFF_TR_SERIAL=1
stats serial_${V}_ecg ecg FDIFF_TR_SERIAL=1 FDIFF_LIB=$VAR
