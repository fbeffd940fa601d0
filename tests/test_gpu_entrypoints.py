"""GPU end-to-end: cmd/train.py -> checkpoint -> cmd/sample.py on a small synthetic dataset (frequency domain),
the flow of SURVEY.md 3.1 / 3.2, plus the datamodule round trips of tests/test_datamodules.py:67-117."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def run(cmd, cwd):
    env = dict(os.environ, PYTHONPATH=str(ROOT))
    r = subprocess.run([sys.executable] + cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r


def test_train_then_sample(tmp_path):
    common = ["fourier_transform=true", "datamodule.max_len=24", "datamodule.num_samples=96",
              "datamodule.n_channels=4", "datamodule.batch_size=32"]
    run([str(ROOT / "cmd" / "train.py"), *common, "score_model.d_model=24", "score_model.num_layers=2",
         "score_model.n_head=4", "trainer.max_epochs=3", "trainer.callbacks.2.every_n_epochs=2",
         "trainer.callbacks.2.num_samples=32", "trainer.callbacks.2.num_diffusion_steps=5", "run_id=testrun"], tmp_path)
    run_dir = tmp_path / "lightning_logs" / "testrun"
    assert (run_dir / "train_config.yaml").exists()
    ckpts = list((run_dir / "checkpoints").glob("epoch=*-val_loss=*.ckpt"))
    assert len(ckpts) == 1
    ck = torch.load(ckpts[0], map_location="cpu", weights_only=False)
    assert "state_dict" in ck and "hyper_parameters" in ck and "pos_encoder.embedding.weight" in ck["state_dict"]
    assert ck["hyper_parameters"]["n_channels"] == 4 and ck["hyper_parameters"]["max_len"] == 24
    run([str(ROOT / "cmd" / "sample.py"), "model_id=testrun", "num_samples=64", "num_diffusion_steps=10",
         "sampler.sample_batch_size=32"], tmp_path)
    X = torch.load(run_dir / "samples.pt")
    assert X.shape == (64, 24, 4) and torch.isfinite(X).all()
    res = yaml.safe_load(open(run_dir / "results.yaml"))
    assert res["num_samples"] == 64
    # the reference's metric collection (cmd/conf/metrics/default.yaml) ran on the engine: time / freq / spectral keys
    for k in ("time_sliced_wasserstein_mean", "freq_sliced_wasserstein_max", "time_marginal_wasserstein_mean_self",
              "freq_marginal_wasserstein_max_dummy", "spectral_marginal_wasserstein_mean"):
        assert k in res and res[k] >= 0.0 and res[k] == res[k], k
    assert len(res["time_sliced_wasserstein_all"]) == 1000 and len(res["time_marginal_wasserstein_all"]) == 24 * 4
    # the same sampling job as two ranks (sharing cuda:0, rendezvous over gloo): each rank samples its shard of the batches with
    # its own Philox counter range, rank 0 gathers on the host, scores and writes
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=str(ROOT), FDIFF_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(ROOT / "cmd" / "sample.py"), "model_id=testrun",
                        "num_samples=128", "num_diffusion_steps=10", "sampler.sample_batch_size=32"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    X2 = torch.load(run_dir / "samples.pt")
    assert X2.shape == (128, 24, 4) and torch.isfinite(X2).all()
    assert not torch.equal(X2[:64], X2[64:])                      # the two ranks drew different noise
    res2 = yaml.safe_load(open(run_dir / "results.yaml"))
    assert res2["num_samples"] == 128 and res2["time_sliced_wasserstein_mean"] >= 0.0


def test_fourier_datamodule_roundtrip_and_standardisation():
    """X == idft(X_tilde) and de-standardise round trip (tests/test_datamodules.py:67-117 of the reference)."""
    from fourierdiffusion_amd.dataloaders.datamodules import TensorDatamodule
    from fourierdiffusion_amd.utils.fourier import destandardize_idft, idft
    g = torch.Generator().manual_seed(42)
    Xtr, Xte = torch.randn(60, 20, 3, generator=g), torch.randn(12, 20, 3, generator=g)
    dm = TensorDatamodule(Xtr, Xte, batch_size=20, fourier_transform=True, standardize=False)
    got = torch.cat([b.X for b in dm.test_dataloader()])
    assert torch.allclose(idft(got).cpu(), Xte, atol=1e-5)
    dm = TensorDatamodule(Xtr, Xte, batch_size=20, fourier_transform=True, standardize=True)
    mean, std = dm.feature_mean_and_std
    val = torch.cat([b.X for b in dm.val_dataloader()])              # standardised with TRAIN statistics
    assert torch.allclose(destandardize_idft(val, mean, std).cpu(), Xte, atol=2e-5)
    torch.manual_seed(0)
    tr = torch.cat([b.X for b in dm.train_dataloader()])
    assert tr.shape == (60, 20, 3) and abs(float(tr.mean())) < 0.05 and abs(float(tr.std()) - 1.0) < 0.05


def _run_bench(extra, launcher):
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FDIFF_BENCH_BACKEND="gloo", PYTHONPATH=root)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    args = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--diffusion-steps", "20"] + extra
    if launcher:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + args
    else:
        cmd = [sys.executable] + args                     # plain `python bench.py --gpus 2`: bench.py launches its own ranks
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("launcher", [False, True])
def test_bench_two_rank_rehearsal(launcher):
    """bench.py's multi-rank path (one process per rank, barrier-bracketed timing, MAX over ranks, rank 0 prints ONE JSON
    line with the whole-job value), driven both ways: plain `python bench.py --gpus 2` (self-launching) and under
    torch.distributed.run.  The box has one GPU, so both ranks share cuda:0 and rendezvous over gloo
    (FDIFF_BENCH_BACKEND); on the 8-GPU node the same code runs with backend "nccl" (= RCCL)."""
    d = _run_bench(["--batch", "64"], launcher)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and "cpu_baseline" not in d
    assert d["metric"] == "sampled series/sec (T=100, C=12)"


def test_bench_train_mode_two_rank_rehearsal():
    """--mode train (BASELINE.json configs[2]): data-parallel optimizer steps, gradient exchange through the GradExchange
    interface (gloo here; fd_allreduce_grads over RCCL on the 8-GPU node)."""
    d = _run_bench(["--mode", "train", "--batch", "8", "--steps", "3"], launcher=False)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["scaling"] == "weak"
    assert d["metric"] == "training series/sec (T=252, C=6)" and d["value"] > 0 and d["dtype"] == "bf16"
    # the roofline names the bracketed training kernel with the largest TOTAL time in the window (fd_prof_end)
    # (all five per-layer kernels are bracketed since round 4)
    assert d["roofline"]["frac"] > 0 and d["roofline"]["kernel"].split(" ")[0] in (
        "k_tr_wgrad", "k_tr_ffn_fwd", "k_tr_ffn_bwd", "k_tr_attn_fwd", "k_tr_attn_bwd")


def test_bench_strong_scaling_rehearsal():
    """--workload mimic --scaling strong (BASELINE.json configs[3]): a fixed total divided over the ranks."""
    d = _run_bench(["--workload", "mimic", "--scaling", "strong", "--batch", "33"], launcher=False)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 33 and d["scaling"] == "strong"
    assert d["metric"] == "sampled series/sec (T=256, C=28)" and d["value"] > 0


def test_bench_default_line_carries_the_secondary_rows():
    """The driver runs `python bench.py` once: the default one-GPU line also reports the training step, the other BASELINE
    shapes (configs[3], configs[4]) and the HBM-bound transforms, each measured by its own process after the headline's timed
    region (`secondary`; FDIFF_BENCH_SECONDARY_QUICK shortens the rows for this test)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FDIFF_BENCH_SECONDARY_QUICK="1", PYTHONPATH=root)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--diffusion-steps", "20", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["metric"] == "sampled series/sec (T=100, C=12)" and d["n_gpus"] == 1
    sec = d["secondary"]
    for k in ("train_nasdaq_T252_B64", "train_ecg_T100_B64", "sample_long_T1024_B64", "sample_mimic_T256_B512", "sample_nasdaq_T252_B512"):
        assert "error" not in sec[k], (k, sec[k])
        assert sec[k]["value"] > 0 and sec[k]["roofline"]["frac"] > 0, (k, sec[k])
    assert sec["train_nasdaq_T252_B64"]["roofline"]["kernel"].startswith("k_tr_")
    assert sec["sample_long_T1024_B64"]["roofline"]["kernel"].startswith(("k_attention_bf16", "k_ffn_ln"))
    assert sec["sample_mimic_T256_B512"]["roofline"]["kernel"].startswith("k_mega")
    hb = sec["hbm_kernels"]["shapes"]
    assert set(hb) == {"B4096_T256_C28", "B512_T1024_C16", "B512_T100_C12"} and all(v["dft_TBps"] > 0.1 for v in hb.values())


def test_ecg_datamodule_preprocessing_on_the_engine(tmp_path):
    """ECGDatamodule (datamodules.py:165-238): CSV layout of the MIT-BIH files, `subsample_localization` keeps the 1000 most
    time-localised series, `smooth_frequency` convolves the spectrum -- both through the engine's spectral utilities."""
    import pandas as pd

    from fourierdiffusion_amd.dataloaders.datamodules import ECGDatamodule
    from oracle import fdiff_oracle as O
    rng = np.random.default_rng(0)
    n_tr, n_te = 1500, 200
    X = rng.normal(size=(n_tr + n_te + 2, 187)).astype(np.float32) * 0.05
    bumps = rng.integers(10, 170, size=X.shape[0])
    for i in range(0, X.shape[0], 2):                       # every other series carries a narrow bump: localised in time
        X[i, bumps[i]: bumps[i] + 5] += 2.0
    y = rng.integers(0, 5, size=(X.shape[0], 1)).astype(np.float32)
    d = tmp_path / "ecg"
    d.mkdir()
    pd.DataFrame(np.concatenate([X[: n_tr + 1], y[: n_tr + 1]], axis=1)).to_csv(d / "mitbih_train.csv", header=False, index=False)
    pd.DataFrame(np.concatenate([X[n_tr + 1:], y[n_tr + 1:]], axis=1)).to_csv(d / "mitbih_test.csv", header=False, index=False)
    dm = ECGDatamodule(data_dir=tmp_path, batch_size=64, fourier_transform=True, standardize=True)
    dm.prepare_data()
    dm.setup()
    assert dm.X_train.shape == (n_tr, 187, 1) and dm.X_test.shape == (n_te, 187, 1) and dm.y_train.dtype == torch.long
    np.testing.assert_allclose(dm.X_train[:, :, 0].numpy(), X[1: n_tr + 1], atol=1e-6)     # first row = header, as in the reference
    sub = ECGDatamodule(data_dir=tmp_path, subsample_localization=True, smooth_frequency=True, smoother_width=2.0)
    sub.setup()
    assert sub.X_train.shape == (1000, 187, 1) and sub.y_train.shape == (1000,)
    # the oracle's ranking picks the same 1000 series; the smoothing is the oracle's
    Xtr = X[1: n_tr + 1, :, None]
    loc, sloc = O.localization_metrics(Xtr)
    keep = np.argsort(loc / sloc, kind="stable")[:1000]
    assert len(set(keep[:700]) - set(np.argsort(loc / sloc)[:1000])) == 0
    want = O.smooth_frequency(Xtr[keep], 2.0)
    got = sub.X_train.cpu().numpy()
    # same set of series (ties aside) -> compare as sets of rows through their sorted norms
    np.testing.assert_allclose(np.sort(np.linalg.norm(got[:, :, 0], axis=1)), np.sort(np.linalg.norm(want[:, :, 0], axis=1)),
                               rtol=1e-4, atol=1e-5)
    with pytest.raises(FileNotFoundError):
        ECGDatamodule(data_dir=tmp_path / "nowhere").prepare_data()
