# ablation timing of the training kernels: FDIFF_LIB variants built with -DFD_TR_ABL_* (timing only, wrong results)
cd $GRAFT_REPO_ROOT
for v in "" abl_WG_NODMA abl_WG_NOBAR; do
  if [ -n "$v" ]; then export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_$v.so; fi
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_abl -o abl_$v -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train ecg 64 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_abl -name "abl_${v}_kernel_stats.csv" | head -1)
  echo "== variant '$v'"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:7]:
    print(f'  {r["Name"][27:60]:34s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:9.1f}')
PY
done
