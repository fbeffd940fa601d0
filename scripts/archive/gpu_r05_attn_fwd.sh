#!/bin/bash
# k_tr_attn_fwd, pipelined unit form (EXPERIMENT, not in the tree: see profiles/r05_train_attn_fwd_keep_bytes.txt item (4); the switch
# FDIFF_TR_ATTN_FWD_NW existed only in that build): parity, then solo kernel time (FDIFF_TR_SERIAL=1) and step time against the
# previous library, with four / eight waves per workgroup
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05_attn_fwd}; mkdir -p $OUT
L=$GRAFT_REPO_ROOT/fourierdiffusion_amd
timeout 900 python -m pytest tests/test_gpu_train_bf16.py tests/test_gpu_benched_shapes.py tests/test_gpu_widths.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
sb() { timeout 120 python scripts/shape_bench.py train $1 64 2>/dev/null | tail -1 | cut -c1-100; }
for rep in 1 2; do
for shp in nasdaq ecg; do
echo "prev      $(FDIFF_LIB=$L/libfdiff_hip_prev.so sb $shp)"
echo "new nw8   $(FDIFF_TR_ATTN_FWD_NW=8 sb $shp)"
echo "new nw4   $(FDIFF_TR_ATTN_FWD_NW=4 sb $shp)"
done
done
for shp in nasdaq ecg; do
for v in prev nw8 nw4; do
  unset FDIFF_LIB FDIFF_TR_ATTN_FWD_NW
  if [ $v = prev ]; then export FDIFF_LIB=$L/libfdiff_hip_prev.so; elif [ $v = nw8 ]; then export FDIFF_TR_ATTN_FWD_NW=8; else export FDIFF_TR_ATTN_FWD_NW=4; fi
  (cd /tmp && export TMPDIR=/tmp && FDIFF_TR_SERIAL=1 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$shp$v -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train $shp 64 > $OUT/$shp$v.log 2>&1)
  echo "$shp $v: $(python scripts/kstats.py $OUT/$shp$v/s_kernel_stats.csv 8 | grep -E 'k_tr_attn_fwd' | cut -c1-60,100-140)"
done
done
