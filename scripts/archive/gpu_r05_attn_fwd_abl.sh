#!/bin/bash
# k_tr_attn_fwd: what the keep-byte loads inside pass 2 cost (timing ablation -DFD_TR_ABL_AF: no load, wrong results), solo times
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_attn_fwd_abl; mkdir -p $OUT
L=$GRAFT_REPO_ROOT/fourierdiffusion_amd
for shp in nasdaq ecg; do
for v in base ${ABLV:-ablaf}; do
  if [ $v = base ]; then unset FDIFF_LIB; else export FDIFF_LIB=$L/libfdiff_hip_$v.so; fi
  (cd /tmp && export TMPDIR=/tmp && FDIFF_TR_SERIAL=1 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$shp$v -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train $shp 64 > $OUT/$shp$v.log 2>&1)
  echo "$shp $v: $(python scripts/kstats.py $OUT/$shp$v/s_kernel_stats.csv 8 | grep -E 'k_tr_attn_fwd' | cut -c1-60,100-140)"
done
done
