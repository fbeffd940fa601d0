#!/bin/bash
# configs[4]: attention staging from the producer's bf16 rows (default) against the fp32 rows (FDIFF_ATT_XROWS=0), alternating on one
# box: 100-step sampler runs, then rocprofv3 per-kernel times of both, then the phase clocks of the rows form.  usage: ... [TAG]
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
F=$OUT/attn_long_xrows_ab.txt
: > $F
for rep in 1 2 3; do
  for x in 0 1; do
    echo "FDIFF_ATT_XROWS=$x: $(FDIFF_ATT_XROWS=$x python scripts/shape_bench.py sample long 64 100 2>&1 | tail -1 | cut -c1-150)" | tee -a $F
  done
done
for x in 0 1; do
  export FDIFF_ATT_XROWS=$x
  echo "-- rocprofv3, FDIFF_ATT_XROWS=$x" | tee -a $F
  bash scripts/attn_abl.sh base 2>&1 | grep -E "k_attention|k_ffn_ln|k_unembed" | cut -c1-200 | tee -a $F
done
unset FDIFF_ATT_XROWS
FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_attnprof.so python scripts/shape_bench.py sample long 64 5 2>&1 | grep -E "attn dbg" | cut -c1-250 | tee $OUT/attn_long_phase_clocks.txt
