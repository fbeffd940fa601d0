#!/bin/bash
# workgroups per CU of the dropout-decision kernel (persistent, VALU-bound, on a side stream beside the forward chain)
sb() { python scripts/shape_bench.py train $1 64 2>/dev/null | tail -1 | cut -c1-110; }
for rep in 1 2; do
for k in 16 8 4 2 1; do
echo "MASK_WGS=$k nasdaq: $(FDIFF_TR_MASK_WGS=$k sb nasdaq)"
echo "MASK_WGS=$k ecg:    $(FDIFF_TR_MASK_WGS=$k sb ecg)"
done
done
