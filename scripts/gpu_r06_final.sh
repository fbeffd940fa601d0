#!/bin/bash
# Round-6 evidence, one box: full GPU suite; the round script (bench line + kernel stats + HBM traffic + training line and stats);
# rocprofv3 of the 1000-step configs[4] run and one --pmc pass of its attention kernel's LDS counters; solo (FDIFF_TR_SERIAL=1) kernel
# times of the training step at both shapes; the training step with the persistent forward off / on (alternating); phase clocks of
# k_tr_fwd_layers when the variant library exists.  usage: bash scripts/gpu_r06_final.sh TAG
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
rm -f $GRAFT_REPO_ROOT/gpurun_out/parity_errors.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/gpu_suite.txt; cat $OUT/gpu_suite.txt
cp $GRAFT_REPO_ROOT/gpurun_out/parity_errors.log $OUT/parity_errors.txt 2>/dev/null
bash scripts/gpu_round.sh $TAG 2>&1 | cut -c1-300 | head -60
cd $GRAFT_REPO_ROOT
bash scripts/gpu_ab.sh $TAG 3 "FDIFF_TR_PERSIST=0" "FDIFF_TR_PERSIST=1" -- python scripts/shape_bench.py train nasdaq 64
bash scripts/gpu_ab.sh $TAG 3 "FDIFF_TR_PERSIST=0" "FDIFF_TR_PERSIST=1" -- python scripts/shape_bench.py train ecg 64
for v in trpprof; do
  if [ -f fourierdiffusion_amd/libfdiff_hip_$v.so ]; then bash scripts/gpu_r06_clk.sh $TAG $v "nasdaq ecg" | cut -c1-700; fi
done
python bench.py --workload long --no-cpu-baseline --no-secondary > $OUT/bench_long.json 2> $OUT/bench_long.err; tail -1 $OUT/bench_long.json | cut -c1-500
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_long -o long -- python $GRAFT_REPO_ROOT/bench.py --workload long --no-cpu-baseline --no-secondary > $OUT/stats_long.log 2>&1)
python scripts/kstats.py $OUT/stats_long/long_kernel_stats.csv 5
rm -f $OUT/stats_long/long_kernel_trace.csv
# ---- LDS counters of the long-series attention kernel (one --pmc pass, no trace domains)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $OUT/pmc_long -o p -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py sample long 64 5 > /dev/null 2>&1)
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/pmc_long/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_attention_bf16" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
with open("$OUT/attn_lds_pmc_summary.txt", "w") as f:
    f.write("# k_attention_bf16<3, ROWS> at T=1024, C=16, B=64 (scripts/shape_bench.py sample long 64 5), one --pmc pass; K image swizzled (round 6)\n")
    for k in sorted(tot): f.write(f"{k:32s} {tot[k]:16.0f}\n")
    if tot.get("SQ_LDS_IDX_ACTIVE"): f.write(f"LDS bank conflict / LDS active       {tot['SQ_LDS_BANK_CONFLICT']/tot['SQ_LDS_IDX_ACTIVE']:.3f}\n")
print(open("$OUT/attn_lds_pmc_summary.txt").read())
PY
rm -rf $OUT/pmc_long
# ---- solo kernel times of the training step
for shp in nasdaq ecg; do
  (cd /tmp && FDIFF_TR_SERIAL=1 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial_$shp -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train $shp 64 > $OUT/serial_$shp.log 2>&1)
  echo "== solo (FDIFF_TR_SERIAL=1) $shp"; python scripts/kstats.py $OUT/serial_$shp/s_kernel_stats.csv 8 | cut -c1-70,100-140
  rm -f $OUT/serial_$shp/s_kernel_trace.csv
done
bash scripts/archive/gpu_r05_shapes.sh $TAG 2>&1 | tail -8
