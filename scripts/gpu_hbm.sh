python -m pytest tests/test_gpu_fourier.py tests/test_gpu_metrics.py tests/test_gpu_entrypoints.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed" | head -20
python scripts/hbm_kernels_bench.py 2>&1 | tail -10 | cut -c1-97
