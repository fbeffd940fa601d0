#!/bin/bash
# A/B of a k_ffn_ln variant at T = 1024 (configs[4] shard): parity tests of the per-layer bf16 path, then 40-step runs + kernel stats
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/ffn_ab; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "bf16 or long or layer or ffn or stepwise or widths" > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
for v in old new old new; do
  if [ $v = old ]; then export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_old.so; else unset FDIFF_LIB; fi
  echo "$v $(timeout 300 python scripts/shape_bench.py sample long 64 40 2>&1 | tail -1)"
done
for v in old new; do
  if [ $v = old ]; then export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_old.so; else unset FDIFF_LIB; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$v -o p -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py sample long 64 20 > /dev/null 2>&1)
  echo "$v"; python scripts/kstats.py $OUT/$v/p_kernel_stats.csv 3 | cut -c1-60,100-170
done
