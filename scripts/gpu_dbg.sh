#!/bin/bash
# debugging helper: run one command, keep head and tail of its combined output
OUT=$GRAFT_REPO_ROOT/gpurun_out/dbg
mkdir -p $OUT
"$@" > $OUT/full.log 2>&1
echo "rc=$?"
head -c 6000 $OUT/full.log
echo ......
tail -c 1500 $OUT/full.log
