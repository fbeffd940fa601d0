"""fourierdiffusion_amd -- MI355X-native engine for the score-matching hot path of
JonathanCrabbe/FourierDiffusion (DFT/iDFT, transformer score network, VE/VP SDE, sampler).

The sub-packages mirror the reference's ``fdiff`` layout (``utils.fourier``,
``schedulers.sde``, ``models.score_models``, ``sampling.sampler`` ...) so that
the hydra ``_target_`` strings keep working through the ``fdiff`` alias package
at the repository root.  All arithmetic runs in hand-written HIP kernels behind
the C ABI of ``include/fdiff_hip.h``; there is no CPU fallback.
"""
import os as _os

# The host driver of this pool only supports dmabuf IPC: RCCL / cross-process device memory need this before the HIP runtime
# starts (it is exported on the GPU boxes already; kept here so that any launcher environment inherits it).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__version__ = "0.1.0"
