#!/bin/bash
# round 4: (1) attention-backward forms test, (2) attention output handed to k_ffn_ln as bf16 rows (default) against fp32 rows
# (FDIFF_ATT_F32ROWS=1): parity tests of the per-layer path, configs[4] A/B.  usage: bash scripts/archive/gpu_r04_att16.sh TAG
TAG=${1:-att16}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_train_bf16.py -m gpu -x -q -k "forms or reproducible or exact_f32" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 1200 python -m pytest tests/test_gpu_sampler_parity_shapes.py tests/test_gpu_benched_shapes.py tests/test_gpu_baseline_shapes.py tests/test_gpu_transformer.py tests/test_gpu_sampler.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert|no tests" | tail -5
grep -h "form" gpurun_out/parity_errors.log | tail -8
for rep in 1 2 3; do
echo "bf16 rows: $(python scripts/shape_bench.py sample long 64 200 2>&1 | tail -1)"
echo "f32 rows:  $(FDIFF_ATT_F32ROWS=1 python scripts/shape_bench.py sample long 64 200 2>&1 | tail -1)"
done
n=long_bf16rows
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py sample long 64 100 > $OUT/$n.log 2>&1)
echo "== $n"; python scripts/kstats.py $OUT/$n/s_kernel_stats.csv 4 | cut -c1-60,100-140
n=long_f32rows
(cd /tmp && export TMPDIR=/tmp && FDIFF_ATT_F32ROWS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py sample long 64 100 > $OUT/$n.log 2>&1)
echo "== $n"; python scripts/kstats.py $OUT/$n/s_kernel_stats.csv 4 | cut -c1-60,100-140
