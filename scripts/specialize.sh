#!/bin/bash
# Ahead-of-time specialisation of the persistent kernel for one workload (the MI355X answer to "no tracing compiler"):
#   scripts/specialize.sh NAME T C [D H L F]        (defaults: the hydra default transformer 72 12 10 2048)
# builds fourierdiffusion_amd/libfdiff_hip_NAME.so with one more static-shape instantiation; run with
#   FDIFF_LIB=$PWD/fourierdiffusion_amd/libfdiff_hip_NAME.so python cmd/sample.py ...
# The workgroup plan (series per workgroup S, head pairs per group NPG, rotation, tiles per wave MT) is the engine's own:
# it is read back from a one-off FDIFF_MEGA_PROF run at the batch size you will use (default 512), so run this on a GPU box.
set -e
NAME=$1; T=$2; C=$3; D=${4:-72}; H=${5:-12}; L=${6:-10}; F=${7:-2048}; B=${FDIFF_SPECIALIZE_BATCH:-512}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PLAN=$(cd $ROOT && FDIFF_MEGA_PROF=1 python - <<PY 2>&1 | grep -o "nw=[0-9]* S=[0-9]* npg=[0-9]* mt=[0-9]* rot=[0-9]*" | head -1
import torch, sys
sys.path.insert(0, "$ROOT")
from fourierdiffusion_amd.models.score_models import ScoreModule
from fourierdiffusion_amd.schedulers.sde import VPScheduler
from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
sch = VPScheduler(fourier_noise_scaling=True); sch.set_noise_scaling($T)
m = ScoreModule(n_channels=$C, max_len=$T, noise_scheduler=sch, d_model=$D, num_layers=$L, n_head=$H).to("cuda")
m.precision = "bf16"; m.eval()
from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
DiffusionSampler(score_model=m, sample_batch_size=$B).sample(num_samples=$B, num_diffusion_steps=2)
PY
)
[ -z "$PLAN" ] && { echo "could not read the plan (is a GPU visible and T <= 256?)"; exit 1; }
S=$(echo $PLAN | sed 's/.* S=\([0-9]*\).*/\1/'); NPG=$(echo $PLAN | sed 's/.*npg=\([0-9]*\).*/\1/')
MT=$(echo $PLAN | sed 's/.*mt=\([0-9]*\).*/\1/'); ROT=$(echo $PLAN | sed 's/.*rot=\([0-9]*\).*/\1/')
KS1=$(( (D + 1 + 31) / 32 )); DT=$(( (D + 15) / 16 )); KSO=$(( (8 * H + 31) / 32 ))
echo "plan: S=$S NPG=$NPG MT=$MT rot=$ROT  tiles: KS1=$KS1 DT=$DT KSO=$KSO"
$ROOT/scripts/build_variant.sh $NAME "-DFD_MEGA_EXTRA_SHAPE=$T,$D,$C,$H,$S,$NPG,$ROT,$L,$F" "-DFD_MEGA_EXTRA_TILES=$KS1,$DT,$KSO,$MT"
