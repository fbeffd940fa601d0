#!/bin/bash
# round-end chain: full GPU suite, smoke, driver-style line, evidence scripts, phase profile + PMC of the persistent kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/full_gpu.log 2>&1; tail -1 gpurun_out/full_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_style.json 2> gpurun_out/bench_driver_style.err; tail -1 gpurun_out/bench_driver_style.json | cut -c1-400
bash scripts/gpu_round.sh r03f 2>&1 | tail -40
bash scripts/gpu_evidence_r03.sh 2>&1 | tail -60
cd $GRAFT_REPO_ROOT
FDIFF_MEGA_PROF=1 python bench.py --steps 1 --warmup 0 --diffusion-steps 40 --no-cpu-baseline --no-secondary 2>&1 | grep "fdiff prof" > gpurun_out/mega_phase_profile.txt; tail -16 gpurun_out/mega_phase_profile.txt
bash scripts/gpu_pmc.sh final > gpurun_out/pmc_final_summary.txt 2>&1; tail -30 gpurun_out/pmc_final_summary.txt
