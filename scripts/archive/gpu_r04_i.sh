#!/bin/bash
# three-way headline A/B: committed library (c60), current tree without the row swap (old), current tree
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r04i; mkdir -p $OUT
for i in 1 2; do for v in c60 old base; do
  if [ $v = base ]; then unset FDIFF_LIB; else export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_$v.so; fi
  echo "$v $(python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("kernel","")[:120])')"
done; done 2>&1 | tee $OUT/ab3.txt
