#!/bin/bash
# k_ffn_ln prologue / epilogue loads unconditional and batched: parity, A/B against the previous library at configs[4] and the
# droughts shape, phase clocks of the new form (-DFD_FFN_PROF variant)
L=$GRAFT_REPO_ROOT/fourierdiffusion_amd
timeout 1200 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_widths.py tests/test_gpu_sampler_parity_shapes.py tests/test_gpu_transformer.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for rep in 1 2 3; do
for a in "long 64" "droughts 512"; do
echo "prev  $(FDIFF_LIB=$L/libfdiff_hip_prev.so timeout 300 python scripts/shape_bench.py sample $a 100 2>&1 | tail -1 | cut -c1-120)"
echo "new   $(timeout 300 python scripts/shape_bench.py sample $a 100 2>&1 | tail -1 | cut -c1-120)"
done
done
for a in "long 64" "droughts 512"; do FDIFF_LIB=$L/libfdiff_hip_ffnprof.so timeout 300 python scripts/shape_bench.py sample $a 10 2>&1 | grep "ffn_ln dbg" | cut -c1-260; done
