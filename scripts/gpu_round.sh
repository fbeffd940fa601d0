#!/bin/bash
# Round-end evidence: full bench line, rocprofv3 kernel stats of the same command, HBM traffic PMC passes; the training
# bench line (--mode train) with its kernel stats.  usage (through gpurun): bash scripts/gpu_round.sh r02
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > $OUT/stats.log 2>&1
python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/stats/bench_kernel_stats.csv 8
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $OUT/pmc_$c.log 2>&1
done
python - <<PY
import csv, glob, json
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = 0.0; n = 0
    for f in glob.glob("$OUT/pmc_%s/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if "k_mega" in r["Kernel_Name"] and r["Counter_Name"] == c:
                tot += float(r["Counter_Value"]); n += 1
    res[c] = (tot, n)
print(res)
if res["FETCH_SIZE"][1]:
    fetch_kb = res["FETCH_SIZE"][0] / res["FETCH_SIZE"][1]; write_kb = res["WRITE_SIZE"][0] / max(1, res["WRITE_SIZE"][1])
    out = {"kernel": "k_mega", "launches": res["FETCH_SIZE"][1], "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
           "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over python bench.py --steps 1 --warmup 0 (one launch = 1000 diffusion steps); FETCH_SIZE doubled (gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md HBM section); units KB. FETCH counts L2 misses, most of them served by the 256 MB MALL: the 6.3 MB of bf16 weights exceed one XCD's 4 MB L2, so every XCD re-fetches them each diffusion step (8 x 6.3 MB x 1000 = 50 GB, plus the positional / embedding tables and x); WRITE = the 2.4 GB of x written back per launch (no register-spill scratch traffic in this build)"}
    json.dump(out, open("$OUT/hbm_traffic.json", "w"), indent=1); print(out)
PY

# ---- training line (BASELINE.json configs[2] per-GPU shard) + per-kernel stats at both training shapes
cd $GRAFT_REPO_ROOT
python bench.py --mode train > $OUT/bench_train.json 2> $OUT/bench_train.err; tail -1 $OUT/bench_train.json | cut -c1-700
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_train -o train_nasdaq -- python $GRAFT_REPO_ROOT/bench.py --mode train --no-cpu-baseline > $OUT/stats_train.log 2>&1
python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/stats_train/train_nasdaq_kernel_stats.csv 8
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_train -o train_ecg -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train ecg 64 > $OUT/stats_train_ecg.log 2>&1
tail -1 $OUT/stats_train_ecg.log
python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT/stats_train/train_ecg_kernel_stats.csv 8
cd $GRAFT_REPO_ROOT
python scripts/hbm_kernels_bench.py > $OUT/hbm_kernels.txt 2>&1; tail -10 $OUT/hbm_kernels.txt
