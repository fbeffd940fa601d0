#!/bin/bash
# k_ffn_ln prologue with every global operand of both owned tiles requested up front: parity of the per-layer path, then
# configs[4] against the previous prologue (libfdiff_hip_old.so), alternating on one box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r04k; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_benched_shapes.py tests/test_gpu_baseline_shapes.py tests/test_gpu_widths.py tests/test_gpu_score.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
  echo "new: $(python scripts/shape_bench.py sample long 64 100 2>&1 | tail -1)"
  echo "old: $(FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_old.so python scripts/shape_bench.py sample long 64 100 2>&1 | tail -1)"
done 2>&1 | tee $OUT/long_ab.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -o long -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py sample long 64 100 > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1
python $GRAFT_REPO_ROOT/scripts/kstats.py $GRAFT_REPO_ROOT/$OUT/stats/long_kernel_stats.csv 4
