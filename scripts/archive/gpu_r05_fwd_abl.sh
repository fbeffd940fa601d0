#!/bin/bash
# k_tr_ffn_fwd: solo kernel time (FDIFF_TR_SERIAL=1, rocprofv3) under the timing ablations of -DFD_TR_ABL_FWD (wrong results):
# 1 no weight DMA in the loop, 2 no barrier, 4 no mask / activity block, 7 all three, 16 no chunk loop (prologue + epilogue only)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05_fwdabl}
mkdir -p $OUT
for shp in ${SHAPES:-nasdaq ecg}; do
for v in ${ABLS:-base 1 2 4 7 16}; do
  if [ $v = base ]; then unset FDIFF_LIB; else export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_fwdabl$v.so; fi
  (cd /tmp && export TMPDIR=/tmp && FDIFF_TR_SERIAL=1 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$shp$v -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train $shp 64 > $OUT/$shp$v.log 2>&1)
  echo "$shp ablation $v: $(python scripts/kstats.py $OUT/$shp$v/s_kernel_stats.csv 8 | grep k_tr_ffn_fwd | cut -c1-60,100-140)"
done
done | tee $OUT/ffn_fwd_ablations.txt
