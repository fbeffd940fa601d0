for e in 4 8 12 16; do
echo "== elements per thread $e"; FDIFF_FFT_EPT=$e python scripts/hbm_kernels_bench.py 2>&1 | tail -10 | cut -c1-97 | sed -n '2,3p;6p'
done
