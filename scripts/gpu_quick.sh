#!/bin/bash
# quick GPU loop: score tests + short bf16 bench + kernel stats (run via gpurun from the repo root)
TAG=${1:-q}
python -m pytest tests/test_gpu_score.py -m gpu -q 2>&1 | tail -4
python bench.py --precision bf16 --steps 1 --warmup 1 --diffusion-steps 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('series/s', round(d['value'],1), 'step_ms', round(d['score_net_step_ms'],4), 'TF', round(d['achieved_tflops_whole_step'],1), d.get('roofline'))"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --precision bf16 --steps 1 --warmup 0 --diffusion-steps 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
python $GRAFT_REPO_ROOT/scripts/kstats.py $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv 8
