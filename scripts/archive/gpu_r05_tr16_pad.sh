#!/bin/bash
# k_tr_wgrad with 13-KiB stage records: step time against the LDS footprint of the launch (FDIFF_TR_WG_LDS_KB pads it; 80 = the
# footprint of the 26-KiB records), previous library alongside
sb() { timeout 120 python scripts/shape_bench.py train $1 64 2>/dev/null | tail -1 | cut -c1-100; }
VAR=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_prev.so
for rep in 1 2 3; do
for shp in nasdaq ecg; do
echo "prev      $(FDIFF_LIB=$VAR sb $shp)"
for kb in 0 56 64 80; do
echo "pad $kb    $(FDIFF_TR_WG_LDS_KB=$kb sb $shp)"
done
done
done
