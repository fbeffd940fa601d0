#!/bin/bash
# in-place vs autosort FFT forms (and elements per thread) at kernel level, after the in-place instantiations lost their spills
for cfg in "0 8" "1 12" "1 16" "1 24" "1 32"; do set -- $cfg
  echo "== FDIFF_FFT_INPLACE=$1 FDIFF_FFT_EPT=$2"
  FDIFF_FFT_INPLACE=$1 FDIFF_FFT_EPT=$2 bash scripts/gpu_hbm_trace.sh 2>&1 | grep k_fft | sed 's/  */ /g'
done
