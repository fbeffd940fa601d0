#!/bin/bash
# per-kernel time of the long-series step under library variants: attn_abl.sh NAME...  ("base" = the regular library)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset FDIFF_LIB; else export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_$v.so; fi
  rm -rf /tmp/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py sample ${SHAPE:-long} ${BATCH:-64} 5 > /tmp/prof_$v.log 2>&1
  echo "== $v slices=${FDIFF_ATTN_SLICES:-auto}: $(grep -i 'ms per' /tmp/prof_$v.log | tail -1)"
  python $GRAFT_REPO_ROOT/scripts/kstats.py $(ls /tmp/prof_$v/*kernel_stats.csv) 4
done
