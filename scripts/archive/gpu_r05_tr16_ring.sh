#!/bin/bash
# k_tr_wgrad with 13-KiB stage records: ring depth 3 / 4 / 5 (the shallower ones padded to the same LDS footprint or not),
# previous library alongside; then solo kernel times
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_tr16_ring; mkdir -p $OUT
L=$GRAFT_REPO_ROOT/fourierdiffusion_amd
sb() { timeout 120 python scripts/shape_bench.py train $1 64 2>/dev/null | tail -1 | cut -c1-100; }
timeout 600 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_benched_shapes.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for rep in 1 2 3; do
for shp in nasdaq ecg; do
echo "prev         $(FDIFF_LIB=$L/libfdiff_hip_prev.so sb $shp)"
echo "nb3 pad 64   $(FDIFF_LIB=$L/libfdiff_hip_nb3.so FDIFF_TR_WG_LDS_KB=64 sb $shp)"
echo "nb4          $(FDIFF_LIB=$L/libfdiff_hip_nb4.so sb $shp)"
echo "nb4 pad 68   $(FDIFF_LIB=$L/libfdiff_hip_nb4.so FDIFF_TR_WG_LDS_KB=68 sb $shp)"
echo "nb5          $(sb $shp)"
done
done
stats() {  # name, shape, env...
  n=$1; shp=$2; shift; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train $shp 64 > $OUT/$n.log 2>&1)
  echo "== $n: $@"; python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/$n/s_kernel_stats.csv 7 | grep wgrad | cut -c1-70,100-140
}
for shp in nasdaq ecg; do
stats s_prev_$shp $shp FDIFF_TR_SERIAL=1 FDIFF_LIB=$L/libfdiff_hip_prev.so
stats s_nb3_$shp $shp FDIFF_TR_SERIAL=1 FDIFF_LIB=$L/libfdiff_hip_nb3.so
stats s_nb4_$shp $shp FDIFF_TR_SERIAL=1 FDIFF_LIB=$L/libfdiff_hip_nb4.so
stats s_nb5_$shp $shp FDIFF_TR_SERIAL=1
done
