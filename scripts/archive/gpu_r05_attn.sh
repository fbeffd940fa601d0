#!/bin/bash
# configs[4]'s attention kernel: query-slice sensitivity (how much of the launch is replicated K/V staging and how much round
# quantisation) and the in-kernel phase clocks (-DFD_ATTN_ABL=3 variant).  usage: bash scripts/archive/gpu_r05_attn.sh [TAG]
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
F=$OUT/attn_long_slices.txt
: > $F
for sl in auto 1 2 4 8; do
  if [ $sl = auto ]; then unset FDIFF_ATTN_SLICES; else export FDIFF_ATTN_SLICES=$sl; fi
  bash scripts/attn_abl.sh base 2>&1 | grep -E "^==|k_attention|k_ffn_ln" | cut -c1-200 >> $F
done
unset FDIFF_ATTN_SLICES
cat $F
FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_attnprof.so python scripts/shape_bench.py sample long 64 5 2>&1 | grep -E "attn dbg|ms per" | cut -c1-250 | tee $OUT/attn_long_phase_clocks.txt
