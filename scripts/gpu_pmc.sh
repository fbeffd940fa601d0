#!/bin/bash
# PMC passes over the persistent kernel (run via gpurun from the repo root).  usage: gpu_pmc.sh TAG [bench args]
TAG=${1:-pmc}; shift
ARGS=${@:---precision bf16 --steps 1 --warmup 0 --diffusion-steps 20 --no-cpu-baseline}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH" \
           "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_mega" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(f"{k:32s} {tot[k]:16.0f}")
w = tot.get("SQ_WAVE_CYCLES", 1)
for k in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_VMEM","SQ_ACTIVE_INST_MISC"):
    if k in tot: print(f"  {k:28s} / WAVE_CYCLES = {tot[k]/w:6.3f}")
if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and "SQ_BUSY_CYCLES" in tot:
    print("  MFMA busy / (4 * BUSY_CU?) raw ratio vs SQ_BUSY_CYCLES:", tot["SQ_VALU_MFMA_BUSY_CYCLES"]/tot["SQ_BUSY_CYCLES"])
PY
