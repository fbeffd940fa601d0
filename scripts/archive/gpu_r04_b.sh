#!/bin/bash
# round 4, call B: FFN-loop schedule variants of k_mega (interleaved sched_group_barrier pipelines, light-wave priority) against
# the round-3 loop, same box; instantiation parity of the new default
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r04b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bench_instantiation.py tests/test_gpu_baseline_shapes.py -m gpu -x -q 2>&1 | tail -3
bash scripts/gpu_variants.sh old base ilv prio old base 2>&1 | tee $OUT/variants.txt | grep -E "==|step_ms"
