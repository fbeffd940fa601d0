#!/bin/bash
# training step (measured with a build that read FDIFF_TR_EVENT_SYS): device-scope event records against system-scope ones; three mask waits in the
# forward (default) against one per layer (FDIFF_TR_MASK_WAIT_ALL=1)
sb() { python scripts/shape_bench.py train $1 64 2>/dev/null | tail -1 | cut -c1-110; }
timeout 900 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_benched_shapes.py tests/test_gpu_widths.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for rep in 1 2 3; do
for cfg in "X=1" "FDIFF_TR_EVENT_SYS=1" "FDIFF_TR_MASK_WAIT_ALL=1" "FDIFF_TR_EVENT_SYS=1 FDIFF_TR_MASK_WAIT_ALL=1"; do
echo "$cfg nasdaq: $(env $cfg python scripts/shape_bench.py train nasdaq 64 2>/dev/null | tail -1 | cut -c40-100)"
echo "$cfg ecg:    $(env $cfg python scripts/shape_bench.py train ecg 64 2>/dev/null | tail -1 | cut -c40-100)"
done
done
