# per-kernel profile of one bf16 optimizer step (B=64): usage  bash scripts/train_prof.sh [ecg|nasdaq]
cd $GRAFT_REPO_ROOT
NAME=${1:-ecg}
python scripts/shape_bench.py train $NAME 64 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -o tr_$NAME -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train $NAME 64 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_train -name "tr_${NAME}*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over the run; per kernel (name, calls, avg us, % of kernel time):")
for r in rows[:24]:
    print(f'  {r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:9.1f} {float(r["Percentage"]):6.2f}')
PY
