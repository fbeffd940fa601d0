#!/bin/bash
# training step with the weight-gradient launches on the chain's own stream (FDIFF_TR_SERIAL=1) against the side-stream form
sb() { python scripts/shape_bench.py train $1 64 2>/dev/null | tail -1 | cut -c1-110; }
for rep in 1 2 3; do
echo "parallel nasdaq: $(sb nasdaq)"
echo "serial   nasdaq: $(FDIFF_TR_SERIAL=1 sb nasdaq)"
echo "parallel ecg:    $(sb ecg)"
echo "serial   ecg:    $(FDIFF_TR_SERIAL=1 sb ecg)"
done
