#!/bin/bash
# Round-5 evidence, one box: full GPU suite; the round script (bench line + kernel stats + HBM traffic + training line and stats);
# rocprofv3 of the 1000-step configs[4] run; PMC passes of k_attention_bf16 (separate --pmc runs, --kernel-trace/--stats only
# elsewhere); solo (FDIFF_TR_SERIAL=1) kernel times of the training step at both shapes.  usage: bash scripts/gpu_r05_final.sh TAG
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
rm -f $GRAFT_REPO_ROOT/gpurun_out/parity_errors.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/gpu_suite.txt; cat $OUT/gpu_suite.txt
cp $GRAFT_REPO_ROOT/gpurun_out/parity_errors.log $OUT/parity_errors.txt 2>/dev/null
bash scripts/gpu_round.sh $TAG 2>&1 | cut -c1-300 | tail -60
cd $GRAFT_REPO_ROOT
python bench.py --workload long --no-cpu-baseline --no-secondary > $OUT/bench_long.json 2> $OUT/bench_long.err; tail -1 $OUT/bench_long.json | cut -c1-500
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_long -o long -- python $GRAFT_REPO_ROOT/bench.py --workload long --no-cpu-baseline --no-secondary > $OUT/stats_long.log 2>&1)
python scripts/kstats.py $OUT/stats_long/long_kernel_stats.csv 5
rm -f $OUT/stats_long/long_kernel_trace.csv
# ---- PMC of the long-series attention kernel
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32" \
           "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_long_$i -o p -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py sample long 64 5 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/pmc_long_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_attention_bf16" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
with open("$OUT/attn_pmc_summary.txt", "w") as f:
    f.write("# k_attention_bf16<3, ROWS> at T=1024, C=16, B=64 (scripts/shape_bench.py sample long 64 5: 5 diffusion steps x 10 layers = 50 launches + warm-up), three --pmc passes\n")
    for k in sorted(tot): f.write(f"{k:32s} {tot[k]:16.0f}\n")
    if tot.get("SQ_INSTS_MFMA"):
        mf = tot["SQ_INSTS_MFMA"]
        f.write(f"non-MFMA VALU per MFMA               {(tot['SQ_INSTS_VALU']-mf)/mf:.2f}\n")
        f.write(f"transcendental share of non-MFMA VALU {tot['SQ_INSTS_VALU_TRANS_F32']/(tot['SQ_INSTS_VALU']-mf):.3f}\n")
        f.write(f"SALU per MFMA                        {tot['SQ_INSTS_SALU']/mf:.2f}\n")
        f.write(f"LDS bank conflict / LDS active       {tot['SQ_LDS_BANK_CONFLICT']/max(1.0,tot['SQ_LDS_IDX_ACTIVE']):.3f}\n")
    if tot.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_VALU_MFMA_BUSY_CYCLES"):
            if k in tot: f.write(f"{k:28s} / WAVE_CYCLES = {tot[k]/tot['SQ_WAVE_CYCLES']:6.3f}\n")
print(open("$OUT/attn_pmc_summary.txt").read())
PY
rm -rf $OUT/pmc_long_*
# ---- solo kernel times of the training step
cd $GRAFT_REPO_ROOT
for shp in nasdaq ecg; do
  (cd /tmp && FDIFF_TR_SERIAL=1 timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial_$shp -o s -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train $shp 64 > $OUT/serial_$shp.log 2>&1)
  echo "== solo (FDIFF_TR_SERIAL=1) $shp"; python scripts/kstats.py $OUT/serial_$shp/s_kernel_stats.csv 8 | cut -c1-70,100-140
  rm -f $OUT/serial_$shp/s_kernel_trace.csv
done
bash scripts/archive/gpu_r05_shapes.sh $TAG 2>&1 | tail -3
