#!/bin/bash
# W1 pair image with row bits 3 <-> 4 swapped (LDS bank slots) against the natural order (libfdiff_hip_old.so), attention-phase
# priority experiments (ap1, ap2); parity of the static shapes first
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r04h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bench_instantiation.py tests/test_gpu_sampler_parity_shapes.py tests/test_gpu_entrypoints.py -m gpu -x -q 2>&1 | tail -3
bash scripts/gpu_ab_headline.sh 2>&1 | head -6 | tee $OUT/ab_swap.txt
for v in ap1 ap2; do
  cp fourierdiffusion_amd/libfdiff_hip_$v.so /tmp/new_$v.so
done
for i in 1 2; do for v in base ap1 ap2; do
  if [ $v = base ]; then unset FDIFF_LIB; else export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_$v.so; fi
  echo "$v $(python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])')"
done; done 2>&1 | tee $OUT/ab_attprio.txt
