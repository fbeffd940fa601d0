#!/bin/bash
# hipcc (library, ROCm 7.2 clang) against hiprtc (torch's bundled ROCm 7.0 image / the system's 7.2 image) on the SAME kernel text:
# the headline ecg shape (FDIFF_MEGA_JIT=force) and the nasdaq shape, alternating on one box.  usage: bash scripts/archive/gpu_r05_jit_ab.sh [TAG]
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export FDIFF_MEGA_JIT_VERBOSE=1
F=$OUT/mega_jit_compiler_ab.txt
: > $F
run() {  # label, env...
  local label=$1; shift
  for w in ecg nasdaq; do
    local args="--no-cpu-baseline --no-secondary --steps 2 --warmup 1"
    [ $w = nasdaq ] && args="$args --workload nasdaq"
    line=$(env "$@" timeout 600 python bench.py $args 2>> $OUT/jit_ab.err | tail -1)
    echo "$label $w: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "series/s", d["ms_per_step"], "ms/launch", "frac", d["roofline"]["frac"])' 2>/dev/null || echo "$line" | cut -c1-200)" | tee -a $F
  done
}
for rep in 1 2; do
  run "hipcc-aot" FDIFF_MEGA_JIT=0
  run "hiprtc-torch-image" FDIFF_MEGA_JIT=force FDIFF_CACHE_DIR=/tmp/jc_torch
  run "hiprtc-rocm-7.2" FDIFF_MEGA_JIT=force FDIFF_CACHE_DIR=/tmp/jc_rocm FDIFF_HIPRTC_LIB=/opt/rocm/lib/libhiprtc.so.7
done
grep -h "compiled in\|unavailable" $OUT/jit_ab.err | sort | uniq -c | cut -c1-300 | tee -a $F
