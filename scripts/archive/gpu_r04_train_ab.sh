#!/bin/bash
# round 4: training step, this session's library against the round's starting one (fourierdiffusion_amd/libfdiff_hip_old.so = commit
# 687aa5a) on ONE box, alternating; what the dropout decisions cost (FDIFF_DROPOUT=0); solo kernel times.  usage: bash scripts/archive/gpu_r04_train_ab.sh TAG
TAG=${1:-trab}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
OLD=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_old.so
sb() { python scripts/shape_bench.py train $1 64 2>/dev/null | tail -1 | cut -c1-110; }
stats() {  # name, shape, env...
  n=$1; shp=$2; shift; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train $shp 64 > $OUT/$n.log 2>&1)
  echo "== $n: $@"; python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/$n/s_kernel_stats.csv ${NK:-8} | cut -c1-70,100-140
}
timeout 900 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_benched_shapes.py tests/test_gpu_widths.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for rep in 1 2 3; do
echo "new nasdaq: $(sb nasdaq)"
echo "old nasdaq: $(FDIFF_LIB=$OLD sb nasdaq)"
echo "new ecg:    $(sb ecg)"
echo "old ecg:    $(FDIFF_LIB=$OLD sb ecg)"
done
echo "new nasdaq p=0: $(FDIFF_DROPOUT=0 sb nasdaq)"
echo "new ecg p=0:    $(FDIFF_DROPOUT=0 sb ecg)"
stats serial_new nasdaq FDIFF_TR_SERIAL=1
stats serial_new_p0 nasdaq FDIFF_TR_SERIAL=1 FDIFF_DROPOUT=0
stats par_new nasdaq FDIFF_X=1
