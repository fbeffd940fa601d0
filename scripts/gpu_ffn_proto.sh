#!/bin/bash
# FFN prototype matrix (scripts/ubench/ffn32_proto.hip): usage  gpu_ffn_proto.sh TAG "flags of variant 1|flags of variant 2|..."
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-ffnproto}
LIST=${2:-"-DILV=0|-DILV=1"}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT/scripts/ubench
: > $OUT/matrix.txt
IFS='|' read -ra VARS <<< "$LIST"
for v in "${VARS[@]}"; do
  hipcc -O3 --offload-arch=gfx950 $v ffn32_proto.hip -o /tmp/ffn32_v 2>/dev/null || { echo "build failed: $v" >> $OUT/matrix.txt; continue; }
  echo "=== $v" >> $OUT/matrix.txt
  timeout 60 /tmp/ffn32_v >> $OUT/matrix.txt 2>&1
done
grep -E "===|256 workgroup|MODE" $OUT/matrix.txt
