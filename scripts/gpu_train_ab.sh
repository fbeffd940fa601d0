#!/bin/bash
# Parametrised driver of the round-6 training-step experiments (replaces gpu_r06_ext / fb / dxg / epi .sh): on ONE box
#   1. the -m gpu tests of the training path (the summary and any assertion text are printed, the log is kept);
#   2. phase clocks, for every variant library named in CLOCK_LIBS (e.g. "fbprof abprof": fourierdiffusion_amd/libfdiff_hip_<name>.so,
#      built with -DFD_TR_PROF_FB / -DFD_TR_PROF_ATTN / -DFD_TRP_PROF -- scripts/gpu_r06_clk.sh prints them);
#   3. the same-box A/B of environment variants (scripts/gpu_ab.sh, three alternating rounds) of scripts/shape_bench.py train at T = 252 and T = 100.
# usage: [CLOCK_LIBS="fbprof"] bash scripts/gpu_train_ab.sh TAG "VAR=a VAR2=b" "VAR=c" ...      (no variant: one unnamed variant = the defaults)
# e.g.   bash scripts/gpu_train_ab.sh dxg "FDIFF_TR_DX_GEMM=0" "FDIFF_TR_DX_GEMM=1"
#        bash scripts/gpu_train_ab.sh ev "FDIFF_TR_EXT_EVENT=0 FDIFF_TR_LEAN_EVENTS=0 FDIFF_TR_EVENT_FENCE=1" "FDIFF_TR_LEAN_EVENTS=0" "A=1"
TAG=${1:?tag}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_train.py tests/test_gpu_train_persist.py tests/test_gpu_benched_shapes.py -x -q -m gpu > $OUT/tests.log 2>&1
grep -E "passed|failed" $OUT/tests.log | tail -2
grep -E "^E  |Error" $OUT/tests.log | head -8 | cut -c1-300
for v in $CLOCK_LIBS; do
  if [ -f fourierdiffusion_amd/libfdiff_hip_$v.so ]; then bash scripts/gpu_r06_clk.sh $TAG $v "nasdaq ecg" | grep -A3 "phase clocks" | cut -c1-420; fi
done
VARS=("$@"); [ ${#VARS[@]} -eq 0 ] && VARS=("A=1")
for shp in nasdaq ecg; do
  bash scripts/gpu_ab.sh $TAG 3 "${VARS[@]}" -- python scripts/shape_bench.py train $shp 64 | cut -c1-170
done
