#!/bin/bash
# host enqueue time vs GPU time of the training step, default and serial (no side streams) forms
mkdir -p gpurun_out/$1
for w in nasdaq ecg; do
  python scripts/host_time.py $w
  FDIFF_TR_SERIAL=1 python scripts/host_time.py $w | sed 's/^/serial /'
done
