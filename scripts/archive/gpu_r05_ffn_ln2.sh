#!/bin/bash
# k_ffn_ln: two tiles per basic block in the prologue (bit 0) / epilogue (bit 1), against the previous library; kernel stats
L=$GRAFT_REPO_ROOT/fourierdiffusion_amd
for rep in 1 2; do
for a in "long 64" "droughts 512" "droughts 2048"; do
echo "prev  $(FDIFF_LIB=$L/libfdiff_hip_prev.so timeout 300 python scripts/shape_bench.py sample $a 60 2>&1 | tail -1 | cut -c1-110)"
for j in 0 1 2; do
echo "j$j    $(FDIFF_LIB=$L/libfdiff_hip_j$j.so timeout 300 python scripts/shape_bench.py sample $a 60 2>&1 | tail -1 | cut -c1-110)"
done
echo "j3    $(timeout 300 python scripts/shape_bench.py sample $a 60 2>&1 | tail -1 | cut -c1-110)"
done
done
