#!/bin/bash
# r03 evidence for the HBM-bound kernels and the T = 1024 path: kernel stats + FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hbm -o hbm -- python $GRAFT_REPO_ROOT/scripts/hbm_kernels_bench.py > $OUT/hbm.log 2>&1
python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/hbm/hbm_kernel_stats.csv 8 | cut -c1-100,100-140
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_hbm_$c -o p -- python $GRAFT_REPO_ROOT/scripts/hbm_kernels_bench.py --json > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/long -o long -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py sample long 64 20 > $OUT/long.log 2>&1
tail -1 $OUT/long.log
python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/long/long_kernel_stats.csv 8 | cut -c1-100,100-140
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32" \
           "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_long_$i -o p -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py sample long 64 5 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
# HBM kernels: traffic per launch at (4096, 256, 28) (grid-size selects the shape: k_fft grid 4096 x 512 threads, k_sde_step ...)
acc = collections.defaultdict(lambda: [0.0, 0])
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/pmc_hbm_%s/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            name = "k_fft_fwd" if "k_fft<false" in k else ("k_fft_inv" if "k_fft<true" in k else ("k_sde_step" if "k_sde_step" in k else None))
            if name and r["Counter_Name"] == c:
                a = acc[(name, r["Grid_Size"] + " / wg " + r["Workgroup_Size"], c)]; a[0] += float(r["Counter_Value"]); a[1] += 1
rows = {}
for (name, grid, c), (tot, n) in acc.items():
    rows.setdefault((name, grid), {})[c] = tot / n
with open("$OUT/hbm_traffic.txt", "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over scripts/hbm_kernels_bench.py --json (shapes (4096,256,28), (512,1024,16), (512,100,12)); MB per launch, FETCH doubled\n")
    f.write("# per the gfx950 note of MI355X_MICROARCH.md; one row per (kernel, grid size = shape)\n")
    for (name, grid), v in sorted(rows.items()):
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            f.write(f"{name:12s} grid {grid:>20s}: fetch {2*v['FETCH_SIZE']/1024:9.1f} MB  write {v['WRITE_SIZE']/1024:9.1f} MB\n")
print(open("$OUT/hbm_traffic.txt").read()[:3000])
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/pmc_long_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_attention_bf16" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
with open("$OUT/attn_pmc_summary.txt", "w") as f:
    f.write("# k_attention_bf16<3> at T=1024, C=16, B=64 (scripts/shape_bench.py sample long 64 5: 5 diffusion steps x 10 layers = 50 launches + warm-up), three --pmc passes\n")
    for k in sorted(tot): f.write(f"{k:32s} {tot[k]:16.0f}\n")
    if tot.get("SQ_INSTS_MFMA"):
        mf = tot["SQ_INSTS_MFMA"]
        f.write(f"non-MFMA VALU per MFMA               {(tot['SQ_INSTS_VALU']-mf)/mf:.2f}\n")
        f.write(f"transcendental share of non-MFMA VALU {tot['SQ_INSTS_VALU_TRANS_F32']/(tot['SQ_INSTS_VALU']-mf):.3f}\n")
        f.write(f"SALU per MFMA                        {tot['SQ_INSTS_SALU']/mf:.2f}\n")
        f.write(f"LDS bank conflict / LDS active       {tot['SQ_LDS_BANK_CONFLICT']/max(1.0,tot['SQ_LDS_IDX_ACTIVE']):.3f}\n")
    if tot.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS"):
            if k in tot: f.write(f"{k:28s} / WAVE_CYCLES = {tot[k]/tot['SQ_WAVE_CYCLES']:6.3f}\n")
print(open("$OUT/attn_pmc_summary.txt").read())
PY
