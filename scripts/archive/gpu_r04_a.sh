#!/bin/bash
# round 4, call A: FFN prototype (32x32x16 H + 16x16x32 W2) against the shipped decomposition, the new benched-shape parity tests,
# the whole GPU suite, and a baseline headline line on the same box
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT/scripts/ubench
hipcc -O3 --offload-arch=gfx950 ffn32_proto.hip -o /tmp/ffn32_proto 2>/dev/null && timeout 120 /tmp/ffn32_proto > $OUT/ffn32_proto.txt 2>&1; cat $OUT/ffn32_proto.txt
hipcc -O3 --offload-arch=gfx950 ffn_pattern.hip -o /tmp/ffn_pattern 2>/dev/null && timeout 120 /tmp/ffn_pattern > $OUT/ffn_pattern.txt 2>&1; tail -14 $OUT/ffn_pattern.txt
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/parity_errors.log
timeout 1500 python -m pytest tests/test_gpu_benched_shapes.py tests/test_gpu_scale_script.py -m gpu -x -q 2>&1 | tail -15
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_benched_shapes.py --deselect tests/test_gpu_scale_script.py 2>&1 | tail -6
cp gpurun_out/parity_errors.log $OUT/parity_errors.log 2>/dev/null
python bench.py --no-secondary --no-cpu-baseline > $OUT/bench_base.json 2> $OUT/bench_base.err; tail -1 $OUT/bench_base.json | cut -c1-400
