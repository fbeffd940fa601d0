#!/bin/bash
# PMC of k_ffn_ln at configs[4] (T = 1024, C = 16, B = 64): separate --pmc passes over scripts/shape_bench.py sample long 64 5
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_ffn_ln_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32" \
           "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py sample long 64 5 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_ffn_ln" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
with open("$OUT/ffn_ln_pmc_summary.txt", "w") as f:
    f.write("# k_ffn_ln<3, 5, 4, 3> at T=1024, C=16, B=64 (scripts/shape_bench.py sample long 64 5: 5 diffusion steps x 10 layers = 50 launches + warm-up), three --pmc passes; sums over the launches\n")
    for k in sorted(tot): f.write(f"{k:32s} {tot[k]:16.0f}   ({n[k]} dispatch records)\n")
    mf = tot.get("SQ_INSTS_MFMA", 0)
    if mf:
        f.write(f"non-MFMA VALU per MFMA               {(tot['SQ_INSTS_VALU']-mf)/mf:.2f}\n")
        f.write(f"LDS instructions per MFMA            {tot['SQ_INSTS_LDS']/mf:.2f}\n")
        f.write(f"SALU per MFMA                        {tot['SQ_INSTS_SALU']/mf:.2f}\n")
        f.write(f"LDS bank conflict / LDS active       {tot['SQ_LDS_BANK_CONFLICT']/max(1.0,tot['SQ_LDS_IDX_ACTIVE']):.3f}\n")
    if tot.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_VALU_MFMA_BUSY_CYCLES"):
            if k in tot: f.write(f"{k:28s} / WAVE_CYCLES = {tot[k]/tot['SQ_WAVE_CYCLES']:6.3f}\n")
    if tot.get("SQ_BUSY_CYCLES") and tot.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        f.write(f"SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = {tot['SQ_VALU_MFMA_BUSY_CYCLES']/tot['SQ_BUSY_CYCLES']:6.3f}\n")
print(open("$OUT/ffn_ln_pmc_summary.txt").read())
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
