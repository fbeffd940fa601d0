"""GPU parity of the Wasserstein evaluation metrics (fdiff.sampling.metrics / fdiff.utils.wasserstein surface) against the
oracle restatement (pinned to the transport LP in tests/test_oracle_golden.py).  Tolerance 2e-5 relative + 2e-6 absolute: the
engine projects in fp32 (the reference projects in float64), the transport itself is exact on the fp32 values."""
import math

import numpy as np
import pytest
import torch

from oracle import fdiff_oracle as O
from oracle import weights as W

from .gpu_util import dev, host

pytestmark = pytest.mark.gpu
RTOL, ATOL = 2e-5, 2e-6


@pytest.mark.parametrize("n,m", [(1000, 1000), (1000, 333), (257, 1024), (64, 1), (1, 64), (2, 3), (5000, 4999)])
def test_w2_sorted_rows_any_sizes(n, m):
    from fourierdiffusion_amd.utils.wasserstein import sort_rows, w2_sorted_rows
    K = 7
    a = W.randn("w2_a", (K, n), 5).astype(np.float32)
    b = (W.randn("w2_b", (K, m), 6) * 1.3 + 0.5).astype(np.float32)
    a[0, : min(n, 10)] = 0.25                                  # ties
    sa, sb = sort_rows(dev(a)), sort_rows(dev(b))
    np.testing.assert_array_equal(host(sa), np.sort(a, axis=1))
    np.testing.assert_array_equal(host(sb), np.sort(b, axis=1))
    got = w2_sorted_rows(sa, sb)
    want = [math.sqrt(O.emd2_1d(a[k], b[k])) for k in range(K)]
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)


def test_sort_rows_negative_zero_and_extremes():
    from fourierdiffusion_amd.utils.wasserstein import sort_rows
    x = np.array([[3.0, -1e30, 0.0, -0.0, 1e-38, -1e-38, 1e30, -2.5]], np.float32)
    np.testing.assert_array_equal(host(sort_rows(dev(x))), np.sort(x, axis=1))


@pytest.mark.parametrize("shape_x,shape_y", [((300, 20, 3), (300, 20, 3)), ((301, 20, 3), (128, 20, 3)), ((50, 187, 1), (64, 187, 1))])
def test_wasserstein_distances_vs_oracle(shape_x, shape_y):
    from fourierdiffusion_amd.utils.wasserstein import WassersteinDistances
    X = W.randn("wd_x", shape_x, 1).astype(np.float32)
    Y = (W.randn("wd_y", shape_y, 2) * 1.2 + 0.1).astype(np.float32)
    wd = WassersteinDistances(original_data=torch.from_numpy(X), other_data=Y, seed=42)
    d = X.shape[1] * X.shape[2]
    dirs = wd.get_random_directions(16)
    np.testing.assert_array_equal(np.stack(dirs), O.random_directions(42, d, 16))          # the reference's draws
    wd = WassersteinDistances(original_data=dev(X), other_data=dev(Y), seed=42)
    np.testing.assert_allclose(wd.sliced_distances(16), O.sliced_distances(X, Y, 42, 16), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(wd.marginal_distances(), O.marginal_distances(X, Y), rtol=RTOL, atol=ATOL)
    assert wd.feature_distance(5) == pytest.approx(O.marginal_distances(X, Y)[5], rel=RTOL, abs=ATOL)
    assert wd.directional_distance(dirs[3]) == pytest.approx(O.sliced_distances(X, Y, 42, 16)[3], rel=RTOL, abs=ATOL)
    std = WassersteinDistances(original_data=X, other_data=Y, normalisation="standardise", seed=42)
    Xf = O.check_flat_array(X)
    np.testing.assert_allclose(std.marginal_distances(), O.marginal_distances(X, Y) / Xf.std(axis=0), rtol=1e-4)
    with pytest.raises(ValueError):
        WassersteinDistances(original_data=X, other_data=Y, normalisation="bogus").marginal_distances()


def test_metric_collection_matches_oracle_and_reference_keys():
    """MetricCollection of cmd/conf/metrics/default.yaml (sliced + marginal, baselines, spectral density) on a small set."""
    from functools import partial

    from fourierdiffusion_amd.sampling.metrics import MarginalWasserstein, MetricCollection, SlicedWasserstein
    X = W.randn("mc_x", (200, 31, 2), 3).astype(np.float32)
    Y = (W.randn("mc_y", (150, 31, 2), 4) * 0.8).astype(np.float32)
    mc = MetricCollection(metrics=[partial(SlicedWasserstein, random_seed=42, num_directions=20, save_all_distances=True),
                                   partial(MarginalWasserstein, random_seed=42, save_all_distances=True)],
                          original_samples=torch.from_numpy(X), include_baselines=True, include_spectral_density=True)
    res = mc(torch.from_numpy(Y))
    assert list(res) == sorted(res)
    Xf, Yf = O.dft(X), O.dft(Y)
    want = {}
    for dom, a, b in (("time", X, Y), ("freq", Xf, Yf)):
        want.update({f"{dom}_{k}": v for k, v in O.sliced_wasserstein_metric(a, b, 42, 20).items()})
        want.update({f"{dom}_{k}": v for k, v in O.marginal_wasserstein_metric(a, b).items()})
    sd = O.marginal_wasserstein_metric(O.spectral_density(X), O.spectral_density(Y), baselines=False)
    want.update({f"spectral_{k}": v for k, v in sd.items()})
    assert sorted(res) == sorted(want)
    for k, v in want.items():
        np.testing.assert_allclose(res[k], v, rtol=2e-4, atol=2e-5, err_msg=k)
    assert isinstance(res["time_sliced_wasserstein_mean"], float) and isinstance(res["time_marginal_wasserstein_all"], list)
    # identical sets: zero distance everywhere
    same = mc(torch.from_numpy(X))
    assert same["time_sliced_wasserstein_max"] == 0.0 and same["freq_marginal_wasserstein_max"] == 0.0


def test_metrics_at_dataset_scale():
    """87k x 187 training set (the reference's ECG size) against 1000 samples, 1000 directions: finite, ordered, and the
    sliced distance to a shifted copy of the set is the length of the shift's projection."""
    from fourierdiffusion_amd.utils.wasserstein import WassersteinDistances
    n, d = 87554, 187
    X = dev(W.randn("big_x", (n, d), 8).astype(np.float32))
    Y = X[:1000] * 1.1
    wd = WassersteinDistances(X, Y, seed=0)
    s = wd.sliced_distances(1000)
    assert s.shape == (1000,) and np.isfinite(s).all() and (s > 0).all()
    shift = torch.full((1, d), 0.5, device=X.device)
    wd2 = WassersteinDistances(X, X + shift, seed=0)
    dirs = np.stack(wd2.get_random_directions(8))
    np.testing.assert_allclose(wd2.directional_distances(dirs), np.abs(dirs.sum(axis=1) * 0.5), rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("shift", [0.0, 0.1, 1.0])
def test_reference_metric_tests(shift):
    """tests/test_metrics.py of the reference (test_sliced_waserstein / test_marginal_waserstein): same data recipe, same
    assertions; POT's sliced estimate is replaced by the oracle's (POT is absent), the marginal ground truth is the shift."""
    from fourierdiffusion_amd.sampling.metrics import MarginalWasserstein, SlicedWasserstein
    np.random.seed(42)
    dataset1 = np.random.rand(1000, 2, 1)
    dataset2 = np.random.rand(1000, 2, 1) + shift
    sw = SlicedWasserstein(original_samples=dataset1, random_seed=42, num_directions=1000, save_all_distances=True)
    metrics = sw(dataset2)
    assert abs(metrics["sliced_wasserstein_mean"] - np.mean(metrics["sliced_wasserstein_all"])) <= 1e-5
    assert metrics["sliced_wasserstein_mean"] <= metrics["sliced_wasserstein_max"]
    want = O.sliced_distances(dataset1, dataset2, 42, 1000)
    np.testing.assert_allclose(metrics["sliced_wasserstein_all"], want, rtol=RTOL, atol=1e-5)
    # POT's sliced_wasserstein_distance is the root of the mean squared directional distance (its own directions): the
    # reference only asks for agreement within 0.1
    assert abs(metrics["sliced_wasserstein_mean"] - math.sqrt(np.mean(want ** 2))) <= 0.1
    mw = MarginalWasserstein(original_samples=dataset1, random_seed=42, save_all_distances=True)
    metrics = mw(dataset2)
    assert abs(metrics["marginal_wasserstein_mean"] - np.mean(metrics["marginal_wasserstein_all"])) <= 1e-5
    assert metrics["marginal_wasserstein_mean"] <= metrics["marginal_wasserstein_max"]
    assert abs(metrics["marginal_wasserstein_mean"] - shift) <= 0.1
    assert abs(metrics["marginal_wasserstein_max"] - shift) <= 0.1
    assert sorted(mw.baseline_metrics) == ["marginal_wasserstein_max_dummy", "marginal_wasserstein_max_self",
                                           "marginal_wasserstein_mean_dummy", "marginal_wasserstein_mean_self"]


def test_c_abi_argument_errors():
    """Every new entry point refuses bad arguments with FD_ERR_ARG and a message (include/fdiff_hip.h conventions)."""
    import ctypes as C

    from fourierdiffusion_amd import _C
    x = torch.zeros(4, 9, 1, device="cuda")
    y = torch.zeros(4, 9, 1, device="cuda")
    h, L = _C.ctx(x.device), _C.lib()
    p, q = x.data_ptr(), y.data_ptr()
    cases = [
        (L.fd_spectral_density(h, None, q, 4, 9, 1, None), b"null"),
        (L.fd_spectral_density(h, p, q, 0, 9, 1, None), b"bad shape"),
        (L.fd_localization_metrics(h, p, p, None, q, 4, 9, 1, None), b"null"),
        (L.fd_frequency_smooth(h, p, 1.0, q, p, 4, 9, 1, None), b"in-place"),
        (L.fd_frequency_smooth(h, p, 0.0, q, q, 4, 9, 1, None), b"sigma"),
        (L.fd_frequency_smooth(h, p, 1.0, q, q, 4, 8, 1, None), b"must be odd"),
        (L.fd_project_rows(h, p, None, q, 4, 9, 1, None), b"null"),
        (L.fd_transpose_rows(h, p, p, 4, 9, None), b"aliased"),
        (L.fd_sort_rows(h, p, p, 4, 9, q, 256, None), b"aliased"),
        (L.fd_sort_rows_temp_bytes(h, 0, 9, C.byref(C.c_size_t(0))), b"bad arguments"),
        (L.fd_w2_sorted_rows(h, p, q, q, 4, 0, 9, None), b"bad shape"),
    ]
    for i, (rc, needle) in enumerate(cases):
        assert rc == -1, (i, rc)
    # the message of the last failure is retrievable
    assert b"bad shape" in L.fd_last_error(h)
    assert L.fd_spectral_density(None, p, q, 4, 9, 1, None) == -1          # no context: error code only
