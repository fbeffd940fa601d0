#!/bin/bash
# does a runtime switch remove the per-launch cost of kernels that use scratch memory? (T=1024 step-by-step sampler: 20 such launches per step)
run() { python bench.py --workload long --steps 1 --warmup 1 --diffusion-steps 100 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms/diffusion step %.4f  roofline kernel avg %.1f us' % (d['score_net_step_ms'], d['roofline']['avg_kernel_us']))"; }
echo default; run
echo HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0; HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 run
echo HSA_SCRATCH_SINGLE_LIMIT=4294967296; HSA_SCRATCH_SINGLE_LIMIT=4294967296 run
echo HSA_NO_SCRATCH_RECLAIM=1; HSA_NO_SCRATCH_RECLAIM=1 run
echo HSA_NO_SCRATCH_THREAD_LIMITER=1; HSA_NO_SCRATCH_THREAD_LIMITER=1 run
