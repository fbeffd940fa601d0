"""Time the spectral utilities of the dataset front-end at the reference's ECG scale (87 554 x 187 x 1) and at the mimic
shape (4096 x 256 x 28); algorithmic bytes / time next to each.  usage: python scripts/spectral_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    from fourierdiffusion_amd.utils.fourier import dft, localization_metrics, smooth_frequency, spectral_density
    g = torch.Generator(device="cuda").manual_seed(0)
    for (B, T, C) in ((87554, 187, 1), (4096, 255, 28)):
        x = torch.randn(B, T, C, device="cuda", generator=g)
        xt = dft(x)
        nbytes = B * T * C * 4
        t = timed(lambda: spectral_density(xt, apply_dft=False))
        print(f"({B},{T},{C}) spectral_density(apply_dft=False): {t * 1e6:.0f} us = {1.5 * nbytes / t / 1e12:.2f} TB/s (read xt, write half)")
        t = timed(lambda: localization_metrics(x))
        print(f"({B},{T},{C}) localization_metrics (dft + O(T^2) per series): {t * 1e6:.0f} us, {2 * B * T * T * 2 / t / 1e12:.2f} TFLOP/s fp32 VALU")
        t = timed(lambda: smooth_frequency(x, 2.0))
        print(f"({B},{T},{C}) smooth_frequency (dft + T x T mixing + idft): {t * 1e6:.0f} us, mixing {2 * B * T * T * C / t / 1e12:.2f} TFLOP/s")


if __name__ == "__main__":
    main()
