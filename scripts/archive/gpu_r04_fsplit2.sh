#!/bin/bash
# F-split of the training FFN kernels at small per-GPU batches (default rule: blocks <= CUs / 4)
sb() { python scripts/shape_bench.py train $1 $2 2>/dev/null | tail -1 | cut -c1-100; }
for cfg in "ecg 8" "ecg 16" "ecg 32" "ecg 40" "nasdaq 8" "nasdaq 16"; do
set -- $cfg
for rep in 1 2; do
echo "split    $1 B=$2: $(sb $1 $2)"
echo "no split $1 B=$2: $(FDIFF_TR_FSPLIT=0 sb $1 $2)"
done
done
