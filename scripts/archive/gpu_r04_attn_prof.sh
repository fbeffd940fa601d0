#!/bin/bash
# in-kernel phase clocks of k_tr_attn_bwd (variant build -DFD_TR_PROF_ATTN, fourierdiffusion_amd/libfdiff_hip_attnprof.so), solo
L=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_attnprof.so
for cfg in "nasdaq X=1" "nasdaq FDIFF_TR_ATTN_OH=0" "ecg X=1"; do
set -- $cfg
echo "== $1 $2 (FDIFF_TR_SERIAL=1)"
env FDIFF_LIB=$L FDIFF_TR_SERIAL=1 $2 python scripts/shape_bench.py train $1 64 2>&1 | grep -E "phase clocks|wave [0-9]:|per optimizer" | cut -c1-220
done
