#!/bin/bash
# LDS / VALU / MFMA activity of the training kernels (one --pmc pass per counter set, no tracing options)
OUT=$GRAFT_REPO_ROOT/gpurun_out/train_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train ${1:-nasdaq} 64 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if k.startswith("k_tr_"):
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("GRBM_GUI_ACTIVE",): n[k] += 1
for k, c in sorted(tot.items()):
    if not c.get("GRBM_GUI_ACTIVE"): continue
    gui = c["GRBM_GUI_ACTIVE"]
    line = f"{k:28s} launches {n[k]:5d} clocks/launch {gui/max(1,n[k]):9.0f}"
    if c.get("SQ_WAVE_CYCLES"):
        line += f" | of wave cycles: VALU {c['SQ_ACTIVE_INST_VALU']/c['SQ_WAVE_CYCLES']:.3f} LDS-inst {c['SQ_ACTIVE_INST_LDS']/c['SQ_WAVE_CYCLES']:.3f} wait-LDS {c['SQ_WAIT_INST_LDS']/c['SQ_WAVE_CYCLES']:.3f} wait-any {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.3f}"
    if c.get("SQ_LDS_IDX_ACTIVE"):
        line += f" | LDS idx active / (256 CUs x clocks) {c['SQ_LDS_IDX_ACTIVE']/ (256.0*gui):.3f} bank conflict/active {c['SQ_LDS_BANK_CONFLICT']/max(1.0,c['SQ_LDS_IDX_ACTIVE']):.3f} | per MFMA: VALU {(c['SQ_INSTS_VALU']-c['SQ_INSTS_MFMA'])/max(1.0,c['SQ_INSTS_MFMA']):.1f} LDS {c['SQ_INSTS_LDS']/max(1.0,c['SQ_INSTS_MFMA']):.2f} SALU {c['SQ_INSTS_SALU']/max(1.0,c['SQ_INSTS_MFMA']):.1f}"
    print(line)
PY
