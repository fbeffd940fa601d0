#!/bin/bash
# same-box A/B of the default bench line: libfdiff_hip_old.so (variant build) against the in-tree library, alternating
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for i in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_old.so; else unset FDIFF_LIB; fi
    echo "$v $(python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])')"
  done
done
unset FDIFF_LIB
for i in 1 2; do python scripts/shape_bench.py sample long 64 40 2>&1 | tail -1; done
