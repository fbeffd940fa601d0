#!/bin/bash
# F-split rule: forward-only split at 100 blocks (mode 1, default) against none (0), both (2), quarter rule for both (3)
sb() { python scripts/shape_bench.py train $1 $2 2>/dev/null | tail -1 | cut -c1-100; }
for rep in 1 2 3; do
for mode in 1 0 2; do
echo "FSPLIT=$mode ecg B=64: $(FDIFF_TR_FSPLIT=$mode sb ecg 64)"
done
done
for mode in 1 3 0; do
echo "FSPLIT=$mode ecg B=48: $(FDIFF_TR_FSPLIT=$mode sb ecg 48)"
echo "FSPLIT=$mode nasdaq B=24: $(FDIFF_TR_FSPLIT=$mode sb nasdaq 24)"
echo "FSPLIT=$mode nasdaq B=32: $(FDIFF_TR_FSPLIT=$mode sb nasdaq 32)"
done
