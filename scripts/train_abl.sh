cd $GRAFT_REPO_ROOT
for v in abl_NOBALLOT abl_NODMA; do
  export FDIFF_LIB=$GRAFT_REPO_ROOT/fourierdiffusion_amd/libfdiff_$v.so
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_abl -o abl_$v -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train ecg 64 > /dev/null 2>&1
  echo "== variant '$v'"; python3 $GRAFT_REPO_ROOT/scripts/kstats.py $(find $GRAFT_REPO_ROOT/gpurun_out/prof_abl -name "abl_${v}_kernel_stats.csv" | head -1) 4 | cut -c1-60,100-140
done
