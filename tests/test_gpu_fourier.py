"""GPU parity: fd_rfft_pack / fd_irfft_unpack (through the fdiff-compatible dft/idft) vs the golden vectors
from the reference and vs the oracle.  Tolerance: 1e-5 abs, the reference's own (tests/test_utils.py:44-51)."""
import numpy as np
import pytest
import torch

from oracle import fdiff_oracle as O
from oracle import weights as W
from oracle.make_golden import DFT_B, DFT_C, DFT_T

from .gpu_util import DEV, dev, host

pytestmark = pytest.mark.gpu
ATOL = 1e-5


@pytest.mark.parametrize("T", DFT_T)
@pytest.mark.parametrize("C", DFT_C)
def test_dft_idft_vs_golden(golden, T, C):
    from fourierdiffusion_amd.utils.fourier import dft, idft
    g = golden("dft")
    x = W.randn(f"dft_x_{T}_{C}", (DFT_B, T, C), 0)
    y = host(dft(dev(x)))
    np.testing.assert_allclose(y, g[f"dft_{T}_{C}"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(y, O.dft(x), atol=ATOL, rtol=0)
    xt = W.randn(f"idft_x_{T}_{C}", (DFT_B, T, C), 0)
    yi = host(idft(dev(xt)))
    np.testing.assert_allclose(yi, g[f"idft_{T}_{C}"], atol=ATOL, rtol=0)


def test_known_answers(golden):
    from fourierdiffusion_amd.utils.fourier import dft
    g = golden("dft")
    for T in (16, 15):
        imp = np.zeros((1, T, 1), np.float32)
        imp[0, 3, 0] = 1.0
        np.testing.assert_allclose(host(dft(dev(imp))), g[f"dft_impulse_{T}"], atol=1e-6)
        n = np.arange(T)
        cosw = np.cos(2 * np.pi * 2 * n / T).astype(np.float32).reshape(1, T, 1)
        np.testing.assert_allclose(host(dft(dev(cosw))), g[f"dft_cos2_{T}"], atol=2e-6)


@pytest.mark.parametrize("shape", [(100, 100, 3), (100, 101, 3),          # the reference's own test_dft shapes
                                   (1, 1, 1), (3, 2, 1), (2, 3, 2), (5, 17, 1), (2, 73, 7), (3, 128, 40),
                                   (2, 1024, 16), (2, 1024, 28), (1, 2048, 3), (7, 365, 9)])
def test_round_trip_both_ways(shape):
    """test_dft of the reference (tests/test_utils.py:36-51): idft(dft(x)) == x and dft(idft(x)) == x."""
    from fourierdiffusion_amd.utils.fourier import dft, idft
    x = dev(W.randn("rt", shape, 11))
    np.testing.assert_allclose(host(idft(dft(x))), host(x), atol=ATOL)
    np.testing.assert_allclose(host(dft(idft(x))), host(x), atol=ATOL)


def test_full_size_properties():
    """BASELINE configs at full size: round trip, linearity and Parseval-with-G (size-independent properties)."""
    from fourierdiffusion_amd.utils.fourier import dft, idft
    for shape in [(1000, 100, 12), (4096, 256, 28), (512, 1024, 16)]:
        a = torch.randn(shape, device=DEV)
        b = torch.randn(shape, device=DEV)
        A, Bt = dft(a), dft(b)
        assert torch.allclose(idft(A), a, atol=2e-5)
        assert torch.allclose(dft(2.5 * a - 0.5 * b), 2.5 * A - 0.5 * Bt, atol=5e-5)
        # energy: ||x||^2 = X0^2 (+ X_{T/2}^2) + 2 * sum of the other packed entries squared
        T = shape[1]
        wgt = torch.full((T,), 2.0, device=DEV)
        wgt[0] = 1.0
        if T % 2 == 0:
            wgt[T // 2] = 1.0
        e_time = (a.double() ** 2).sum(dim=1)
        e_freq = ((A.double() ** 2) * wgt[None, :, None]).sum(dim=1)
        assert torch.allclose(e_time, e_freq, rtol=1e-4)


def test_cpu_tensor_round_trips_through_the_gpu():
    """cmd/sample.py:82 hands idft a CPU tensor; the engine computes on the GPU and returns on the CPU."""
    from fourierdiffusion_amd.utils.fourier import dft, idft
    x = torch.randn(4, 50, 3)
    y = dft(x)
    assert y.device.type == "cpu" and y.shape == x.shape and not y.requires_grad
    np.testing.assert_allclose(y.numpy(), O.dft(x.numpy()), atol=ATOL)
    assert torch.allclose(idft(y), x, atol=ATOL)


def test_fused_standardisation(golden):
    from fourierdiffusion_amd.utils.fourier import destandardize_idft, dft_standardize
    g = golden("dataset")
    X = W.randn("ds_x", (16, 24, 3), 5)
    mean, std = dev(g["mean"]), dev(g["std"])
    Xs = dft_standardize(dev(X), mean, std)
    np.testing.assert_allclose(host(Xs)[3], g["item3"], atol=2e-5)
    np.testing.assert_allclose(host(destandardize_idft(Xs, mean, std)), X, atol=2e-5)


def test_argument_errors():
    from fourierdiffusion_amd import _C
    from fourierdiffusion_amd.utils.fourier import dft
    with pytest.raises(AssertionError):
        dft(torch.zeros(4, 5, device=DEV))
    x = torch.zeros(2, 8, 2, device=DEV)
    h = _C.ctx(x.device)
    rc = _C.lib().fd_rfft_pack(h, x.data_ptr(), x.data_ptr(), 2, 8, 2, None)     # in place is refused
    assert rc == -1 and b"in-place" in _C.lib().fd_last_error(h)


# ---------------------------------------------------------------- spectral utilities (fourier.py:90-209)
@pytest.mark.parametrize("T", (16, 100, 101, 187))
@pytest.mark.parametrize("C", (1, 12))
def test_spectral_utilities_vs_golden(golden, T, C):
    from fourierdiffusion_amd.utils.fourier import localization_metrics, smooth_frequency, spectral_density
    g = golden("spectral")
    x = W.randn(f"spec_x_{T}_{C}", (3, T, C), 0)
    x[1, T // 3: T // 3 + 4] += 3.0
    d = host(spectral_density(dev(x)))
    assert d.shape == (3, T // 2 + 1, C)
    np.testing.assert_allclose(d, g[f"dens_{T}_{C}"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(d, O.spectral_density(x), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(host(spectral_density(dev(x), apply_dft=False)), g[f"dens_nodft_{T}_{C}"], rtol=1e-5, atol=1e-6)
    loc, sloc = localization_metrics(dev(x))
    assert loc.shape == (3,) and sloc.shape == (3,)
    np.testing.assert_allclose(np.stack([host(loc), host(sloc)]), g[f"loc_{T}_{C}"], rtol=1e-4)
    if T % 2 == 1:
        for sigma in (1.0, 4.5):
            y = host(smooth_frequency(dev(x), sigma))
            np.testing.assert_allclose(y, g[f"smooth_{T}_{C}_{sigma}"], atol=2e-5)
            np.testing.assert_allclose(y, O.smooth_frequency(x, sigma), atol=2e-5)
    else:
        with pytest.raises(RuntimeError):                    # like the reference (its einsum raises for even lengths)
            smooth_frequency(dev(x), 1.0)


def test_spectral_utilities_properties_at_dataset_size():
    """ECG-sized front-end call (20000 x 187 x 1): Parseval for the density, white noise is delocalised in both domains,
    the sigma -> 0 limit of the smoothing; CPU inputs come back on the CPU."""
    from fourierdiffusion_amd.utils.fourier import dft, localization_metrics, smooth_frequency, spectral_density
    B, T, C = 20000, 187, 1
    x = dev(W.randn("spec_big", (B, T, C), 2))
    d = spectral_density(x)
    two_sided = d.sum(dim=1) + d[:, 1:].sum(dim=1)
    np.testing.assert_allclose(host(two_sided), host((x * x).sum(dim=1)), rtol=2e-4)
    loc, sloc = localization_metrics(x)
    assert torch.isfinite(loc).all() and torch.isfinite(sloc).all()
    # white noise is delocalised in both domains: a little below the uniform-energy value sum_d d^2 / T (the min over
    # the centre s picks the most favourable one), never above it
    dd = np.minimum(np.arange(T), T - np.arange(T)).astype(np.float64)
    uniform = float((dd**2).sum() / T)
    for v in (loc, sloc):
        assert 0.75 * uniform < float(v.mean()) < uniform and float(v.max()) <= uniform * (1 + 1e-4)
    np.testing.assert_allclose(host(loc[:16]), O.localization_metrics(host(x[:16]))[0], rtol=1e-4)
    # sigma -> 0 is NOT the identity in the reference: Re X_k and Im X_k share a frequency, so each becomes their mean
    ys = dft(smooth_frequency(x[:64], 0.05))
    xt = dft(x[:64])
    n_real, K = T // 2 + 1, (T - 1) // 2
    mean_k = 0.5 * (xt[:, 1:1 + K] + xt[:, n_real:])
    np.testing.assert_allclose(host(ys[:, 0]), host(xt[:, 0]), atol=2e-5)
    np.testing.assert_allclose(host(ys[:, 1:1 + K]), host(mean_k), atol=2e-5)
    np.testing.assert_allclose(host(ys[:, n_real:]), host(mean_k), atol=2e-5)
    np.testing.assert_allclose(host(smooth_frequency(x[:8], 2.0)), O.smooth_frequency(host(x[:8]), 2.0), atol=2e-5)
    xc = torch.from_numpy(W.randn("spec_cpu", (2, 101, 3), 3))
    out = spectral_density(xc)
    assert out.device.type == "cpu" and out.shape == (2, 51, 3)


@pytest.mark.parametrize("B,T", [(37, 187), (4, 100), (5, 16), (130, 255), (2000, 187), (33, 1024)])
def test_single_channel_sets_batched_path(B, T):
    """C == 1 (the reference's ECG layout, (n, 187, 1)): several series share a workgroup; parity with the oracle for odd group
    tails, both directions, and the fused standardise variants."""
    from fourierdiffusion_amd.utils.fourier import destandardize_idft, dft, dft_standardize, idft
    x = W.randn("sc_x", (B, T, 1), 21)
    xt = W.randn("sc_xt", (B, T, 1), 22)
    np.testing.assert_allclose(host(dft(dev(x))), O.dft(x), atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(idft(dev(xt))), O.idft(xt), atol=ATOL, rtol=0)
    mean = W.randn("sc_m", (T, 1), 23)
    std = np.abs(W.randn("sc_s", (T, 1), 24)) + 0.5
    got = host(dft_standardize(dev(x), dev(mean), dev(std)))
    np.testing.assert_allclose(got, (O.dft(x) - mean) / std, atol=3e-5, rtol=0)
    back = host(destandardize_idft(dev(got), dev(mean), dev(std)))
    np.testing.assert_allclose(back, x, atol=3e-5, rtol=0)


@pytest.mark.parametrize("B,T,C", [(5, 256, 28), (3, 1024, 16), (7, 100, 12), (4, 252, 8), (6, 255, 4), (3, 64, 40)])
def test_sixteen_byte_passes_with_standardisation(B, T, C):
    """C % 4 == 0: the load / store passes move two channel pairs per 16-byte access, also through the (T, C) mean / std tables
    of the fused standardise variants (in place for radix-16 lengths, autosort otherwise): parity with the oracle, both ways."""
    from fourierdiffusion_amd.utils.fourier import destandardize_idft, dft, dft_standardize, idft
    x = W.randn("v4_x", (B, T, C), 31)
    xt = W.randn("v4_xt", (B, T, C), 32)
    mean = W.randn("v4_m", (T, C), 33)
    std = np.abs(W.randn("v4_s", (T, C), 34)) + 0.5
    np.testing.assert_allclose(host(dft(dev(x))), O.dft(x), atol=ATOL, rtol=0)
    np.testing.assert_allclose(host(idft(dev(xt))), O.idft(xt), atol=ATOL, rtol=0)
    got = host(dft_standardize(dev(x), dev(mean), dev(std)))
    np.testing.assert_allclose(got, (O.dft(x) - mean) / std, atol=3e-5, rtol=0)
    back = host(destandardize_idft(dev(xt), dev(mean), dev(std)))
    np.testing.assert_allclose(back, O.idft(xt * std + mean), atol=3e-5, rtol=0)


def test_multichannel_smoothing_across_scratch_chunks():
    """(4096, 255, 28): the mixing runs as transpose -> fp32-MFMA GEMM -> transpose over chunks of series sized by the context's
    GEMM scratch (2 chunks here); series on both sides of the chunk boundary match the oracle."""
    from fourierdiffusion_amd.utils.fourier import smooth_frequency
    B, T, C = 4096, 255, 28
    x = W.randn("mc_smooth", (B, T, C), 31)
    y = smooth_frequency(dev(x), 3.0)
    assert y.shape == (B, T, C)
    pick = [0, 1, 2348, 2349, 2350, 4095]
    np.testing.assert_allclose(host(y[pick]), O.smooth_frequency(x[pick], 3.0), atol=3e-5)


@pytest.mark.parametrize("mode", ["dense", "fft"])
def test_transform_paths_agree_with_golden(golden, mode, monkeypatch):
    """Lengths dominated by a large prime factor run as one dense (T, T) fp32-MFMA matrix product, the others as Stockham
    stages; FDIFF_DFT forces either path for every length, and both must match the reference's vectors."""
    from fourierdiffusion_amd.utils.fourier import dft, idft
    monkeypatch.setenv("FDIFF_DFT", mode)
    g = golden("dft")
    for T in DFT_T:
        for C in DFT_C:
            x = W.randn(f"dft_x_{T}_{C}", (DFT_B, T, C), 0)
            np.testing.assert_allclose(host(dft(dev(x))), g[f"dft_{T}_{C}"], atol=ATOL, rtol=0, err_msg=f"dft {mode} {T} {C}")
            xt = W.randn(f"idft_x_{T}_{C}", (DFT_B, T, C), 0)
            np.testing.assert_allclose(host(idft(dev(xt))), g[f"idft_{T}_{C}"], atol=ATOL, rtol=0, err_msg=f"idft {mode} {T} {C}")
    for shape in ((37, 187, 1), (300, 251, 12), (2500, 134, 10), (9, 67, 3)):      # chunked / single-channel / prime lengths
        x = W.randn("paths", shape, 41)
        np.testing.assert_allclose(host(dft(dev(x))), O.dft(x), atol=ATOL, rtol=0)
        np.testing.assert_allclose(host(idft(dev(x))), O.idft(x), atol=ATOL, rtol=0)


@pytest.mark.parametrize("inplace", ["1", "0"])
def test_inplace_and_autosort_forms_agree_with_the_oracle(inplace, monkeypatch):
    _forms(inplace, monkeypatch)


def _forms(inplace, monkeypatch):
    """Smooth lengths (prime factors 2, 3, 5, 7) run the in-place decimation-in-frequency form (one LDS image + a
    digit-reversal table: four workgroups per CU where the Stockham ping-pong fits two); FDIFF_FFT_INPLACE=0 forces the
    autosort form.  Both against the oracle at mixed radices, odd lengths, channel chunks, single-channel (batched) sets, a
    ragged last group and the fused standardisation."""
    from fourierdiffusion_amd.utils.fourier import dft, idft
    monkeypatch.setenv("FDIFF_FFT_INPLACE", inplace)
    monkeypatch.setenv("FDIFF_DFT", "fft")
    shapes = [(5, 100, 12), (3, 252, 6), (4, 256, 28), (2, 1024, 16), (3, 360, 5), (2, 2048, 3), (7, 105, 4), (3, 81, 2), (2, 125, 7),
              (3, 343, 3), (2, 1000, 9), (300, 64, 1), (37, 100, 1), (9, 7, 3), (6, 2, 2), (4, 4096, 2), (3, 256, 70), (2, 1024, 40)]
    for shape in shapes:
        x = W.randn(f"inpl_{shape}", shape, 43)
        np.testing.assert_allclose(host(dft(dev(x))), O.dft(x), atol=ATOL, rtol=0, err_msg=f"dft {shape} inplace={inplace}")
        np.testing.assert_allclose(host(idft(dev(x))), O.idft(x), atol=ATOL, rtol=0, err_msg=f"idft {shape} inplace={inplace}")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(64, 256, 28, generator=g).to(DEV)
    assert torch.allclose(idft(dft(x)).to(DEV), x, atol=2e-5)
