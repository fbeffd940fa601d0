#!/bin/bash
# k_attention_bf16: last partial key block padded to a whole one (FDIFF_ATTN_PAD_MIN: 9 = never, default 3): parity, then
# sampling at the droughts shape (T = 365: 23 -> 24 tiles)
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for rep in 1 2; do
for pm in 9 3; do
echo "pad_min $pm  $(FDIFF_ATTN_PAD_MIN=$pm timeout 300 python scripts/shape_bench.py sample droughts 512 100 2>&1 | tail -1 | cut -c1-150)"
done
done
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_attn_pad; mkdir -p $OUT
for pm in 9 3; do
(cd /tmp && export TMPDIR=/tmp && FDIFF_ATTN_PAD_MIN=$pm timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pm$pm -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py sample droughts 512 100 > $OUT/pm$pm.log 2>&1)
echo "== pad_min $pm"; python scripts/kstats.py $OUT/pm$pm/s_kernel_stats.csv 3 | cut -c1-70,100-140
done
