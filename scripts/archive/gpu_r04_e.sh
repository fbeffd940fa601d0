#!/bin/bash
# headline A/B (libfdiff_hip_old.so vs the in-tree library), then cycles per step over a long run for both (FDIFF_MEGA_PROF)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r04e; mkdir -p $OUT
bash scripts/gpu_ab_headline.sh 2>&1 | head -6 | tee $OUT/ab.txt
for v in old base; do
  if [ $v = base ]; then unset FDIFF_LIB; else export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_$v.so; fi
  echo "== $v"; FDIFF_MEGA_PROF=1 python bench.py --precision bf16 --steps 1 --warmup 0 --diffusion-steps 400 --no-cpu-baseline --no-secondary 2>&1 | grep -E "fdiff prof.*cycles/step|score_net_step_ms" | cut -c1-200
done 2>&1 | tee $OUT/prof_long.txt
rocm-smi --showclocks 2>/dev/null | head -20
