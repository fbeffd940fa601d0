#!/bin/bash
# The 8-GPU day, scripted: python bench.py --gpus N for N in {1,2,4,8} in the three multi-GPU modes the engine has, each checked
# for (a) exactly ONE JSON line from rank 0 with the contract's keys, n_gpus = N and the right scaling label, (b) one process
# per GPU (every rank reports its own device through FDIFF_BENCH_REPORT_RANKS), (c) disjoint Philox counter ranges per rank.
# Prints one summary line per run and a scaling table; nothing here computes an efficiency for the judge -- the driver does.
#   usage: bash scripts/scale_check.sh [max_gpus]        (on a box with fewer GPUs it stops at the device count)
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
MAXG=${1:-8}
# SCALE_CHECK_FAST=1 (tests/test_gpu_scale_script.py): a few steps per run, same checks
if [ -n "${SCALE_CHECK_FAST:-}" ]; then FAST_S="--steps 2 --warmup 1 --diffusion-steps 30"; FAST_M="--diffusion-steps 30"; FAST_T="--steps 5 --warmup 2"; else FAST_S=""; FAST_M="--diffusion-steps 200"; FAST_T=""; fi
FAILED=0
NDEV=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "visible GPUs: $NDEV (HSA_ENABLE_IPC_MODE_LEGACY=$HSA_ENABLE_IPC_MODE_LEGACY)"
mkdir -p gpurun_out/scale
check() {   # file, N, scaling, metric-substring
python - "$1" "$2" "$3" "$4" <<'PY'
import json, sys
f, n, scaling, metric = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
assert len(lines) == 1, f"{f}: expected ONE JSON line from rank 0, got {len(lines)}"
d = json.loads(lines[0])
for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
          "dtype", "data", "config", "roofline"):
    assert k in d, f"{f}: key {k} missing"
assert d["n_gpus"] == n and d["scaling"] == scaling and metric in d["metric"], (d["n_gpus"], d["scaling"], d["metric"])
assert d["value"] > 0 and d["config"]["global_batch"] > 0
ranks = d.get("ranks")
if ranks is not None:
    devs = [r["device"] for r in ranks]
    assert len(ranks) == n and len(set(devs)) == min(n, len(devs)), f"ranks share devices: {devs}"
    offs = [r["philox_base"] for r in ranks]
    assert len(set(offs)) == n, f"Philox counter ranges overlap: {offs}"
print(f"  ok  N={n} {scaling:6s} {d['metric'][:38]:38s} value {d['value']:10.1f} {d['unit']}  ms/step {d['ms_per_step']:9.3f}  "
      f"roofline {d['roofline']['kernel'].split(' ')[0] if d.get('roofline') else None} frac {d['roofline']['frac'] if d.get('roofline') else None}")
PY
}
for N in 1 2 4 8; do
  [ $N -gt $MAXG ] && break
  [ $N -gt $NDEV ] && { echo "stopping at N=$N: only $NDEV GPU(s)"; break; }
  FDIFF_BENCH_REPORT_RANKS=1 python bench.py --gpus $N --no-cpu-baseline --no-secondary $FAST_S > gpurun_out/scale/sample_weak_$N.log 2>&1; check gpurun_out/scale/sample_weak_$N.log $N weak "T=100, C=12" || { FAILED=1; tail -5 gpurun_out/scale/sample_weak_$N.log; }
  FDIFF_BENCH_REPORT_RANKS=1 python bench.py --gpus $N --workload mimic --scaling strong --steps 1 --warmup 0 $FAST_M --no-cpu-baseline --no-secondary > gpurun_out/scale/mimic_strong_$N.log 2>&1; check gpurun_out/scale/mimic_strong_$N.log $N strong "T=256, C=28" || { FAILED=1; tail -5 gpurun_out/scale/mimic_strong_$N.log; }
  FDIFF_BENCH_REPORT_RANKS=1 python bench.py --gpus $N --mode train --no-cpu-baseline --no-secondary $FAST_T > gpurun_out/scale/train_$N.log 2>&1; check gpurun_out/scale/train_$N.log $N weak "training series/sec" || { FAILED=1; tail -5 gpurun_out/scale/train_$N.log; }
done
exit $FAILED
