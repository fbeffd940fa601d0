#!/bin/bash
# token splits of k_tr_wgrad after this round's changes to it
sb() { python scripts/shape_bench.py train $1 64 2>/dev/null | tail -1 | cut -c40-100; }
for rep in 1 2; do
for ts in 12 16 20 24 32; do
echo "TS=$ts nasdaq: $(FDIFF_TR_TS=$ts sb nasdaq)   ecg: $(FDIFF_TR_TS=$ts sb ecg)"
done
done
