python -m pytest tests/test_gpu_fourier.py tests/test_gpu_sde.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for kb in 72 40 24 140; do
echo "== FDIFF_FFT_LDS_KB=$kb"; FDIFF_FFT_LDS_KB=$kb python scripts/hbm_kernels_bench.py 2>&1 | tail -10 | cut -c1-112
done
