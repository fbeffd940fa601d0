"""fourierdiffusion_amd -- MI355X-native engine for the score-matching hot path of
JonathanCrabbe/FourierDiffusion (DFT/iDFT, transformer score network, VE/VP SDE, sampler).

The sub-packages mirror the reference's ``fdiff`` layout (``utils.fourier``,
``schedulers.sde``, ``models.score_models``, ``sampling.sampler`` ...) so that
the hydra ``_target_`` strings keep working through the ``fdiff`` alias package
at the repository root.  All arithmetic runs in hand-written HIP kernels behind
the C ABI of ``include/fdiff_hip.h``; there is no CPU fallback.
"""
__version__ = "0.1.0"
