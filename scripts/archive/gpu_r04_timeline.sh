#!/bin/bash
# kernel time line of one optimizer step (rocprofv3 --kernel-trace), T=100 B=64: start / end / duration / queue per kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline
mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python $GRAFT_REPO_ROOT/scripts/shape_bench.py train ${1:-ecg} 64 > $OUT/tl.log 2>&1)
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/tl_kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_adamw' in r['Kernel_Name']]
a,b=idx[10],idx[11]
t0=int(rows[a]['End_Timestamp'])
for r in rows[a+1:b+1]:
    st=int(r['Start_Timestamp'])-t0; en=int(r['End_Timestamp'])-t0
    name=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:40]
    print(f"{st/1000:8.1f} {en/1000:8.1f} {(en-st)/1000:6.1f} q{r['Queue_Id']:>2s} {name}")
PY
