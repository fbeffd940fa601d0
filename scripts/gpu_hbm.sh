python -m pytest tests/test_gpu_fourier.py tests/test_gpu_sde.py tests/test_gpu_metrics.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
echo "== prefetch"; python scripts/hbm_kernels_bench.py 2>&1 | tail -10 | cut -c1-112
echo "== prefetch, grid x2"; FDIFF_FFT_WGMULT=2 python scripts/hbm_kernels_bench.py 2>&1 | tail -10 | cut -c1-112
echo "== no prefetch (r02 form)"; FDIFF_FFT_NOPREFETCH=1 python scripts/hbm_kernels_bench.py 2>&1 | tail -10 | cut -c1-112
