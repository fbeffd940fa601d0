echo "== full"; python scripts/hbm_kernels_bench.py 2>&1 | tail -10 | cut -c1-112 | head -3
echo "== no stages (load + LDS round trip + store)"; FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_hip_fft_nostage.so python scripts/hbm_kernels_bench.py 2>&1 | tail -10 | cut -c1-112 | head -3
cd /tmp && export TMPDIR=/tmp
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fftpmc -o p -- python $GRAFT_REPO_ROOT/scripts/hbm_kernels_bench.py > /dev/null 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/fftpmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_fft" in r["Kernel_Name"] and r["Grid_Size"] in ("2097152",):     # 4096 workgroups x 512 threads
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(r["Kernel_Name"][:60], r["Counter_Name"])] += 1
for k, v in acc.items():
    print(k, {c: round(x / n[(k, c)]) for c, x in v.items()})
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/fftpmc
done
