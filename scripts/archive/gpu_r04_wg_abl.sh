#!/bin/bash
# k_tr_wgrad ablation builds (wrong results, timing only): what bounds the FFN-role block loop?  Solo times (FDIFF_TR_SERIAL=1).
OUT=$GRAFT_REPO_ROOT/gpurun_out/wgabl
mkdir -p $OUT
for v in "" _wg_NOBAR _wg_NODMA _wg_NOMASK _wg_NOT; do
  n=lib$v
  (cd /tmp && export TMPDIR=/tmp && FDIFF_TR_SERIAL=1 FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train nasdaq 64 > $OUT/$n.log 2>&1)
  echo "== variant '$v'"; python3 $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/$n/s_kernel_stats.csv 8 | grep -E "wgrad" | cut -c1-60,100-140
done
