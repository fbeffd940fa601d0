#!/bin/bash
# profile ablation builds of the library: usage gpu_abl.sh NAME1 NAME2 ...
for v in "$@"; do
  export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_$v.so
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl_$v -o a -- python $GRAFT_REPO_ROOT/bench.py --precision bf16 --steps 1 --warmup 0 --diffusion-steps 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/abl_$v.log 2>&1)
  echo "== $v"; python $GRAFT_REPO_ROOT/scripts/kstats.py $GRAFT_REPO_ROOT/gpurun_out/abl_$v/a_kernel_stats.csv 4 | grep -E "ffn|attn|gemm"
done
