#!/bin/bash
# time library variants: usage gpu_variants.sh NAME...   ("base" = the regular library)
for v in "$@"; do
  if [ "$v" = base ]; then unset FDIFF_LIB; else export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_$v.so; fi
  echo "== $v (FDIFF_MEGA_DBG=${FDIFF_MEGA_DBG:-0})"; python bench.py --precision bf16 --steps 1 --warmup 1 --diffusion-steps 100 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_ms', round(d['score_net_step_ms'],4))"
  FDIFF_MEGA_PROF=1 python bench.py --precision bf16 --steps 1 --warmup 0 --diffusion-steps 8 --no-cpu-baseline 2>&1 | grep "fdiff prof" | grep -v "workgroup\|step [0-9]\|group 1\|wave [1-35-7]" | cut -c13-80
done
