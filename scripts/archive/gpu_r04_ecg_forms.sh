#!/bin/bash
# T=100 (7 token tiles): attention-backward form and waves per workgroup after this round's changes
sb() { python scripts/shape_bench.py train ecg 64 2>/dev/null | tail -1 | cut -c40-100; }
for rep in 1 2 3; do
echo "pair NW=8 (default): $(sb)"
echo "pair NW=4:           $(FDIFF_TR_ATTN_NW=4 sb)"
echo "one head, bf16 parts: $(FDIFF_TR_ATTN_OH=2 sb)"
echo "one head, fp32 parts: $(FDIFF_TR_ATTN_OH=1 sb)"
done
