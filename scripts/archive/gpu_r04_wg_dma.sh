#!/bin/bash
# k_tr_wgrad: where a block issues the staging DMA of block + 2 (0 = behind the barrier, 1 = behind the H / d H MFMAs, 2 = at the end)
OUT=$GRAFT_REPO_ROOT/gpurun_out/wgdma
mkdir -p $OUT
sb() { python scripts/shape_bench.py train $1 64 2>/dev/null | tail -1 | cut -c1-110; }
for rep in 1 2; do
for v in "" _wgdma1 _wgdma2; do
echo "lib$v nasdaq: $(FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip$v.so sb nasdaq)"
echo "lib$v ecg:    $(FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip$v.so sb ecg)"
done
done
for v in "" _wgdma1 _wgdma2; do
  n=lib$v
  (cd /tmp && export TMPDIR=/tmp && FDIFF_TR_SERIAL=1 FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train nasdaq 64 > $OUT/$n.log 2>&1)
  echo "== variant '$v'"; python3 $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/$n/s_kernel_stats.csv 8 | grep -E "wgrad" | cut -c1-60,100-140
done
