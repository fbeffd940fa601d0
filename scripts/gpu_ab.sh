#!/bin/bash
# Parametrised same-box A/B driver (replaces the one-off scripts/gpu_r0[45]_*.sh): alternates environment variants of ONE command on
# one box, ROUNDS times, and prints the command's last line per run.
#   usage: bash scripts/gpu_ab.sh TAG ROUNDS "VAR=a VAR2=b" "VAR=c" ... -- command...
# e.g.   bash scripts/gpu_ab.sh trp 3 "FDIFF_TR_PERSIST=0" "FDIFF_TR_PERSIST=1" -- python scripts/shape_bench.py train nasdaq 64
TAG=$1; ROUNDS=$2; shift 2
VARIANTS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do VARIANTS+=("$1"); shift; done
shift
OUT=${GRAFT_REPO_ROOT:-.}/gpurun_out/$TAG
mkdir -p $OUT
echo "# $(date -u +%FT%TZ) A/B on one box: $*" | tee -a $OUT/ab.txt
for r in $(seq 1 $ROUNDS); do
  for v in "${VARIANTS[@]}"; do
    line=$(env $v timeout 600 "$@" 2>&1 | tail -1 | cut -c1-400)
    echo "round $r [$v] $line" | tee -a $OUT/ab.txt
  done
done
