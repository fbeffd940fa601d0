#!/bin/bash
# per-step time against the number of diffusion steps, two library variants (same box)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r04d; mkdir -p $OUT
for n in 8 20 50 100 300; do for v in ${VARIANTS:-f16 base}; do
  if [ $v = base ]; then unset FDIFF_LIB; else export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_$v.so; fi
  echo "n=$n $v $(FDIFF_MEGA_DBG=${DBG:-0} python bench.py --precision bf16 --steps 2 --warmup 1 --diffusion-steps $n --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_ms', round(d['score_net_step_ms'],4))")"
done; done 2>&1 | tee $OUT/steps_${DBG:-0}.txt
