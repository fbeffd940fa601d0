#!/bin/bash
# the -m gpu suite with its summary kept under gpurun_out/ (gpurun returns only the tail of stdout): bash scripts/gpu_suite.sh TAG [pytest args]
TAG=${1:-suite}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu "$@" > $OUT/pytest.log 2>&1
grep -E "passed|failed|error" $OUT/pytest.log | tail -5 | tee $OUT/summary.txt
