"""GPU parity: SDE kernels (fd_sde_step, fd_prior_sample, fd_perturb, fd_dsm_loss, fd_randn) vs golden vectors
from the reference and vs the oracle.  Tolerances: scheduler step 1e-6 relative/abs at O(1)."""
import numpy as np
import pytest
import torch

from oracle import fdiff_oracle as O
from oracle import weights as W
from oracle.make_golden import SDE_CASES

from .gpu_util import DEV, dev, host, oracle_sde

pytestmark = pytest.mark.gpu


def make(kind, p, scaling, T):
    from fourierdiffusion_amd.schedulers.sde import VEScheduler, VPScheduler
    s = (VPScheduler(beta_min=p[0], beta_max=p[1], fourier_noise_scaling=scaling) if kind == "vp"
         else VEScheduler(sigma_min=p[0], sigma_max=p[1], fourier_noise_scaling=scaling))
    s.set_noise_scaling(T)
    return s


def test_marginal_step_prior_vs_golden(golden):
    g = golden("sde")
    B, T, C = 4, 20, 3
    x = W.randn("sde_x", (B, T, C), 1)
    score = W.randn("sde_score", (B, T, C), 1)
    z = W.randn("sde_z", (B, T, C), 1)
    tvals = np.array([1e-5, 0.1, 0.5, 1.0], np.float32)
    for ci, (kind, p) in enumerate(SDE_CASES):
        for scaling in (False, True):
            tag = f"{kind}{ci}_{int(scaling)}"
            s = make(kind, p, scaling, T)
            mean, std = s.marginal_prob(dev(x), dev(tvals))
            np.testing.assert_allclose(host(mean), g[f"mean_{tag}"], rtol=2e-6, atol=1e-7)
            np.testing.assert_allclose(host(std)[1:], g[f"std_{tag}"][1:], rtol=5e-6, atol=1e-9)
            # VP std at t=1e-5 is sqrt(1 - exp(-2e-6..)) in float32: catastrophic cancellation in the reference too
            np.testing.assert_allclose(host(std)[0], g[f"std_{tag}"][0], rtol=5e-2 if kind == "vp" else 5e-6)
            s.set_timesteps(1000)
            for ti, tv in enumerate((0.37, 1e-5, 1.0)):
                o = s.step(dev(score), tv, dev(x), noise=dev(z)).prev_sample
                np.testing.assert_allclose(host(o), g[f"step_{tag}_{ti}"], rtol=2e-6, atol=2e-6)
                osde = oracle_sde(kind, p, scaling, T)
                np.testing.assert_allclose(host(o), O.sde_step(osde, score, tv, x, z, float(s.step_size)),
                                           rtol=2e-6, atol=2e-6)
            pr = s.prior_sampling((B, T, C), noise=dev(z))
            np.testing.assert_allclose(host(pr), g[f"prior_{tag}"], rtol=1e-6, atol=1e-7)


def test_reference_shape_tests():
    """tests/test_schedulers.py:35-66 of the reference: add_noise and step keep the shape (VE & VP)."""
    from fourierdiffusion_amd.schedulers.sde import VEScheduler, VPScheduler
    for cls in (VEScheduler, VPScheduler):
        s = cls()
        x = torch.randn(50, 20, 3, device=DEV)
        noise = torch.randn(50, 20, 3, device=DEV)
        t = torch.rand(50, device=DEV)
        assert s.add_noise(original_samples=x, noise=noise, timesteps=t).shape == x.shape
        s = cls()
        s.set_noise_scaling(max_len=20)
        s.set_timesteps(num_diffusion_steps=1000)
        out = s.step(torch.randn(50, 20, 3, device=DEV), timestep=0.5, sample=noise)
        assert out.prev_sample.shape == noise.shape and torch.isfinite(out.prev_sample).all()


def test_philox_normals_are_standard_and_reproducible():
    from fourierdiffusion_amd import _C
    n = 1 << 22
    a = torch.empty(n, device=DEV)
    b = torch.empty(n, device=DEV)
    h = _C.ctx(a.device)
    L = _C.lib()
    _C.check(L.fd_randn(h, a.data_ptr(), n, 42, 0, None), h)
    _C.check(L.fd_randn(h, b.data_ptr(), n, 42, 0, None), h)
    torch.cuda.synchronize()
    assert torch.equal(a, b)                                   # counter-based: same (seed, offset) -> same bits
    _C.check(L.fd_randn(h, b.data_ptr(), n // 2, 42, n // 8, None), h)
    torch.cuda.synchronize()
    assert torch.equal(a[n // 2:], b[: n // 2])                # offset addressing: element e <-> counter e/4
    _C.check(L.fd_randn(h, b.data_ptr(), n, 43, 0, None), h)
    assert not torch.equal(a, b)
    ad = a.double()
    assert abs(ad.mean().item()) < 4 / np.sqrt(n)
    assert abs(ad.var().item() - 1.0) < 5e-3
    assert abs((ad ** 3).mean().item()) < 2e-2
    assert abs((ad ** 4).mean().item() - 3.0) < 5e-2
    assert (a.abs() > 4.0).float().mean().item() < 2e-4 and a.abs().max().item() < 7.0
    # odd length exercises the ragged tail
    c = torch.full((1003,), 9.0, device=DEV)
    _C.check(L.fd_randn(h, c.data_ptr(), 1001, 42, 0, None), h)
    torch.cuda.synchronize()
    assert torch.equal(c[:1001], a[:1001]) and c[1001] == 9.0 and c[1002] == 9.0


def test_step_with_device_noise_matches_injected_noise():
    """The on-device Philox path must equal the injected-noise path fed with the same normals."""
    from fourierdiffusion_amd import _C, _rng
    s = make("vp", (0.1, 20.0), True, 100)
    s.set_timesteps(1000)
    x = torch.randn(8, 100, 12, device=DEV)
    score = torch.randn(8, 100, 12, device=DEV)
    torch.manual_seed(7)
    out = s.step(score, 0.4, x).prev_sample
    torch.manual_seed(7)
    key, off0 = _rng.stream()               # the same key the step drew from torch's generator
    z = torch.empty_like(x)
    h = _C.ctx(x.device)
    _C.check(_C.lib().fd_randn(h, z.data_ptr(), z.numel(), key, off0, None), h)
    out2 = s.step(score, 0.4, x, noise=z).prev_sample
    assert torch.equal(out, out2)


def test_step_full_size_linearity():
    """At BASELINE size (512,100,12): the step is affine in (x, score, z) -- check against a float64 evaluation."""
    s = make("vp", (0.1, 20.0), True, 100)
    s.set_timesteps(1000)
    x, sc, z = (torch.randn(512, 100, 12, device=DEV) for _ in range(3))
    out = s.step(sc, 0.73, x, noise=z).prev_sample
    osde = oracle_sde("vp", (0.1, 20.0), True, 100)
    ref = O.sde_step(osde, host(sc), 0.73, host(x), host(z), float(s.step_size))
    np.testing.assert_allclose(host(out), ref, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("lw", [False, True])
def test_perturb_and_loss_vs_oracle(lw):
    from fourierdiffusion_amd import _C
    B, T, C = 5, 37, 3
    X = W.randn("pl_x", (B, T, C), 8)
    z = W.randn("pl_z", (B, T, C), 8)
    t = W.uniform("pl_t", (B,), 8, 0.05, 1.0)
    score = W.randn("pl_s", (B, T, C), 8)
    for kind, p in SDE_CASES[:2]:
        s = make(kind, p, True, T)
        osde = oracle_sde(kind, p, True, T)
        xn, target, std = s.perturb(dev(X), dev(t), noise=dev(z))
        oxn, otg, ostd = O.perturb(osde, X, t, z)
        np.testing.assert_allclose(host(xn), oxn, rtol=5e-6, atol=5e-6)
        np.testing.assert_allclose(host(std), ostd, rtol=1e-5)
        np.testing.assert_allclose(host(target), otg, rtol=2e-5, atol=1e-5)
        loss = torch.empty(1, device=DEV)
        dscore = torch.empty(B, T, C, device=DEV)
        h = _C.ctx(loss.device)
        sd = dev(score)
        _C.check(_C.lib().fd_dsm_loss(h, sd.data_ptr(), target.data_ptr(), std.data_ptr(), int(lw),
                                      loss.data_ptr(), dscore.data_ptr(), B, T, C, None), h)
        want = O.dsm_loss(score, host(target), host(std), lw)
        np.testing.assert_allclose(loss.item(), want, rtol=2e-6)
        # gradient by central differences on a few coordinates of the oracle loss
        hs = host(dscore)
        for (b, k, c) in [(0, 0, 0), (2, 17, 1), (4, 36, 2)]:
            e = np.zeros_like(score, dtype=np.float64)
            e[b, k, c] = 1e-3
            fd = (O.dsm_loss(score + e, host(target), host(std), lw) -
                  O.dsm_loss(score - e, host(target), host(std), lw)) / 2e-3
            np.testing.assert_allclose(hs[b, k, c], fd, rtol=1e-4, atol=1e-7)


def test_generator_and_dropout_decisions_bit_exact_vs_oracle():
    """Integer work, bit-exact bar: the device Philox4x32-10 (fd_philox_words) equals the oracle's restatement -- which the CPU
    suite pins to the published known-answer vectors -- for counters that cross the 32-bit boundary, and the 16 dropout
    decisions per evaluation of the bf16 training path (fd_dropout_decisions) equal the oracle's rule at several p."""
    import ctypes as C

    import torch

    from fourierdiffusion_amd import _C
    from oracle import fdiff_oracle as O
    lib = _C.lib()
    dev = torch.device("cuda", 0)
    ctx = _C.ctx(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    n = 100_003
    for seed, offset in [(0, 0), (0x1234567890ABCDEF, 0xFFFFFFF0), (0xFFFFFFFFFFFFFFFF, (1 << 40) + 12345)]:
        words = torch.empty(n, 4, dtype=torch.int32, device=dev)
        _C.check(lib.fd_philox_words(ctx, words.data_ptr(), n, seed, offset, st), ctx)
        ref = O.engine_philox_words(seed, offset, n)
        got = words.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, ref), (seed, offset)
        for p in (0.1, 0.5, 0.003, 0.9):
            dec = torch.empty(n, dtype=torch.int16, device=dev)
            _C.check(lib.fd_dropout_decisions(ctx, dec.data_ptr(), n, C.c_float(p), seed, offset, st), ctx)
            assert np.array_equal(dec.cpu().numpy().view(np.uint16), O.dropout_decisions16(ref, p)), (seed, offset, p)
    assert tuple(int(v) for v in O.engine_philox_words(0, 0, 1)[0]) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)

