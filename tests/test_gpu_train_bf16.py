"""GPU parity of the bf16 MFMA training path (csrc/fd_train_bf16.hip): forward with dropout + backward in five fused
kernels per encoder layer.  Anchors:
  * the reference's autograd gradients (tests/golden/loss.npz: dropout 0, injected t and z) for the small models,
  * the engine's exact-f32 backward (itself pinned to those golden gradients at 2e-4) for the default transformer and the
    BASELINE training shape (nasdaq T=252, C=6),
at bf16 tolerances, stated per test and logged with the measured values (profiles/r02_parity_errors.txt):
  * every tensor: ||g - ref|| <= 3e-2 ||ref|| and max |g - ref| <= 8e-2 max |ref| for d_model >= 60 (measured: 0.4-2.5e-2
    and 0.5-7e-2); the toy models (d_model 8 / 24: 8- and 24-term dot products, linear1 gradients of 1e-6) l2 <= 8e-2 and NO
    per-tensor max bound (one relu flip is 0.44 of a few-hundred-element tensor's maximum: a bound that passes asserts nothing);
  * whole gradient: cosine with the reference >= 0.9995, ||g - ref|| <= 2e-2 ||ref||;
  * loss (bf16 forward) within 1e-2 relative.
linear1.weight/bias carry the largest error of all tensors by construction: relu' is discontinuous, so a hidden unit whose
pre-activation changes sign under bf16 rounding (|h| below ~0.4 % of its scale) contributes an O(1) relative error for that
(token, unit) -- every mixed-precision backward has these; all other tensors sit at 0.3-1 %.
Plus bit-reproducibility (no float atomics), dropout forward/backward consistency by a central difference, and a short
optimisation run.
"""
import numpy as np
import pytest
import torch

from oracle import weights as W
from oracle.make_golden import CFG_DEFAULT, CFG_ODD, CFG_TINY, SDE_CASES

from .gpu_util import DEV, dev, host, make_model

pytestmark = pytest.mark.gpu
CFGS = {"tiny": CFG_TINY, "odd": CFG_ODD}


def batch_of(X, t):
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    return DiffusableBatch(X=dev(X), y=None, timesteps=dev(t))


def _log(line):
    import os
    print(line)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        with open(os.path.join(root, "gpurun_out", "parity_errors.log"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def _compare_grads(tag, got, ref, max_tol=8e-2, l2_tol=3e-2):
    """got/ref: dict name -> numpy; every tensor must meet both bounds, and the whole gradient the global ones."""
    rows = []
    for k, r in ref.items():
        g = got[k]
        sc = max(np.abs(r).max(), 1e-20)
        rows.append((np.abs(g - r).max() / sc, np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-20), k))
    wm = max(rows, key=lambda x: x[0])
    wl = max(rows, key=lambda x: x[1])
    gf = np.concatenate([got[k].ravel() for k in ref])
    rf = np.concatenate([np.asarray(ref[k], dtype=np.float64).ravel() for k in ref])
    cos = float(gf @ rf / (np.linalg.norm(gf) * np.linalg.norm(rf)))
    glob = float(np.linalg.norm(gf - rf) / np.linalg.norm(rf))
    _log(f"[parity] bf16 training gradients {tag}: worst max-rel {wm[0]:.3e} ({wm[2]}), worst l2-rel {wl[1]:.3e} ({wl[2]}), "
         f"whole gradient: cosine {cos:.6f}, l2-rel {glob:.3e}")
    # max_tol None: no per-tensor max bound (toy widths, where it would have to be ~0.5 and assert nothing) -- per-tensor l2 + cosine only
    bad = [(k, f"{a:.3e}", f"{b:.3e}") for a, b, k in rows if not ((max_tol is None or a <= max_tol) and b <= l2_tol)]
    if bad:
        for a, b, k in sorted(rows, reverse=True)[:8]:
            g, r = got[k], ref[k]
            idx = np.unravel_index(np.abs(g - r).argmax(), r.shape)
            print(f"   {k}: max-rel {a:.3e} l2-rel {b:.3e} at {idx}: got {g[idx]:.4e} ref {r[idx]:.4e} (max |ref| {np.abs(r).max():.4e})")
    assert not bad, (tag, bad)
    assert cos >= 0.9995 and glob <= 2e-2 + (l2_tol - 3e-2), (tag, cos, glob)


def _grads_of(m):
    return {k: host(v) for k, v in m.grad_views().items() if k != "time_encoder.W"}


@pytest.mark.parametrize("name,B", [("tiny", 5), ("odd", 3)])
def test_loss_and_gradients_vs_reference_autograd(golden, name, B):
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    g = golden("loss")
    cfg = CFGS[name]
    X = W.randn(f"loss_x_{name}", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn(f"loss_z_{name}", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform(f"loss_t_{name}", (B,), 3, 0.05, 1.0)
    for ci, (kind, p) in enumerate(SDE_CASES[:2]):
        tag = f"{name}_{kind}{ci}_0"
        m, sch, _ = make_model(cfg, kind, p, precision="bf16")
        m.dropout = 0.0
        m.zero_grad()
        tr = get_sde_loss_fn(sch, train=True)(m, batch_of(X, t), noise=dev(z))
        assert m.train_mode_effective == "bf16"
        assert abs(tr.item() - g[f"loss_train_{tag}"]) <= 1e-2 * abs(g[f"loss_train_{tag}"]), (tr.item(), g[f"loss_train_{tag}"])
        ref = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(f"grad_{tag}/")}
        got = _grads_of(m)
        assert float(m.grad_views()["time_encoder.W"].abs().max()) == 0.0
        # d_model = 8 / 24 (toys): 8- and 24-term dot products, one bf16 rounding (2^-9) is a larger share of every sum, and a
        # tensor has a few hundred elements: a single relu flip moves ONE element of linear1.weight by up to 0.44 of the tensor
        # maximum (measured, logged).  A per-tensor max bound loose enough to pass (0.5) asserts nothing, so these cases assert
        # what does hold them: every tensor's l2-rel <= 8e-2 and the whole gradient's cosine >= 0.9995 / l2-rel <= 7e-2.
        _compare_grads(tag, got, ref, max_tol=None, l2_tol=8e-2)


SHAPES = {
    "default_ecg": (dict(CFG_DEFAULT), 6),
    "nasdaq": (dict(T=252, C=6, D=72, L=10, H=12), 3),
    "class_default": (dict(T=50, C=3, D=60, L=3, H=12), 5),
    "ragged_T": (dict(T=37, C=5, D=72, L=2, H=12), 7),
    "droughts_T365": (dict(T=365, C=3, D=72, L=2, H=12), 2),     # 23 token tiles (odd, ragged): one-head attention backward
}


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_gradients_vs_exact_f32_engine(name):
    """Default transformer (d_model 72, 10 layers, ff 2048) and the BASELINE training shape: bf16 backward against the
    exact-f32 backward of the same engine (dropout 0, same t and z)."""
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    cfg, B = SHAPES[name]
    X = W.randn(f"tb_x_{name}", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn(f"tb_z_{name}", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform(f"tb_t_{name}", (B,), 3, 0.05, 1.0)
    res = {}
    for prec in ("fp32", "bf16"):
        m, sch, _ = make_model(cfg, precision=prec)
        m.dropout = 0.0
        m.zero_grad()
        loss = get_sde_loss_fn(sch, train=True)(m, batch_of(X, t), noise=dev(z)).item()
        assert m.train_mode_effective == prec
        res[prec] = (loss, _grads_of(m))
    assert abs(res["bf16"][0] - res["fp32"][0]) <= 1e-2 * abs(res["fp32"][0]), (res["bf16"][0], res["fp32"][0])
    _compare_grads(name, res["bf16"][1], res["fp32"][1])


@pytest.mark.parametrize("cfg,B,p", [(dict(T=37, C=5, D=72, L=2, H=12), 7, 0.0), (dict(T=200, C=4, D=72, L=2, H=12), 3, 0.1),
                                      (dict(T=50, C=3, D=60, L=3, H=12), 5, 0.1), (dict(T=100, C=2, D=24, L=2, H=3), 4, 0.1)])
def test_attention_backward_forms_agree(monkeypatch, cfg, B, p):
    """k_tr_attn_bwd per (head pair, series) (FDIFF_TR_ATTN_OH=0), per (head, series) with fp32 partial tensors of d x (=1) and
    with bf16 ones (=2; the default from 12 token tiles on): same Philox key, dropout on / off, ragged T, an odd head count.
    Form 1 computes every gradient with the pair form's arithmetic except that a pair's two heads enter d x as two fp32 terms
    instead of one MFMA accumulation: a last-bit difference in d x that flips single bf16 operand roundings downstream
    (measured: <= 3.1e-4 of the tensor maximum; bound 2e-3); form 2 rounds the per-head d x contributions to bf16 (measured
    <= 2.4e-3, bound 5e-3; the comparison against the exact-f32 engine at the benched shapes is tests/test_gpu_benched_shapes.py); all three
    are bit-reproducible."""
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    tag = f"T{cfg['T']}_D{cfg['D']}_H{cfg['H']}_p{p}"
    X = W.randn(f"oh_x_{tag}", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn(f"oh_z_{tag}", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform(f"oh_t_{tag}", (B,), 3, 0.05, 1.0)
    m, sch, _ = make_model(cfg, precision="bf16")
    m.dropout = p
    fn = get_sde_loss_fn(sch, train=True)
    res = {}
    for form in ("0", "1", "2", "2"):
        monkeypatch.setenv("FDIFF_TR_ATTN_OH", form)
        m.zero_grad()
        torch.manual_seed(77)
        loss = fn(m, batch_of(X, t), noise=dev(z)).item()
        assert m.train_mode_effective == "bf16"
        g = m.grads.clone()
        if form in res:
            assert loss == res[form][0] and torch.equal(g, res[form][1]), "one-head form is not bit-reproducible"
        res[form] = (loss, g, _grads_of(m))
    assert res["0"][0] == res["1"][0] == res["2"][0]               # the forward does not depend on the backward form
    for form, tol in (("1", 2e-3), ("2", 5e-3)):
        worst = max(np.abs(res[form][2][k] - r).max() / max(np.abs(r).max(), 1e-20) for k, r in res["0"][2].items())
        _log(f"[parity] attention backward form {form} vs pair form ({tag}): worst max-rel over the tensors {worst:.3e}")
        assert worst <= tol, (form, worst)


@pytest.mark.parametrize("cfg,B,p", [(dict(T=37, C=5, D=72, L=2, H=12), 7, 0.0), (dict(T=252, C=6, D=72, L=3, H=12), 3, 0.1),
                                      (dict(T=50, C=3, D=60, L=3, H=12), 5, 0.1), (dict(T=100, C=2, D=24, L=2, H=3), 4, 0.1),
                                      (dict(T=20, C=3, D=8, L=2, H=4), 6, 0.1)])
def test_attention_input_gradient_as_one_product_agrees_with_the_partial_tensors(monkeypatch, cfg, B, p):
    """Round 6: k_tr_attn_bwd writes the d(q | k | v) rows and the CONSUMER of d x (the next k_tr_ffn_bwd's prologue; k_tr_dx0 behind
    layer 0) multiplies them by in_proj^T with fp32 accumulation over all heads (default), instead of one partial tensor of d x per
    head / head pair (FDIFF_TR_DX_GEMM=0).  Against the partial-tensor form with fp32 parts (the pair form: the exact one) the
    difference is the order of fp32 additions in d x, which flips single bf16 operand roundings downstream; both are bit-reproducible.
    Covers ragged T, a length that is not a multiple of 16 at 16 tiles (one head per workgroup), an odd head count, head_dim 2."""
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    tag = f"T{cfg['T']}_D{cfg['D']}_H{cfg['H']}_p{p}"
    X = W.randn(f"dxg_x_{tag}", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn(f"dxg_z_{tag}", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform(f"dxg_t_{tag}", (B,), 3, 0.05, 1.0)
    m, sch, _ = make_model(cfg, precision="bf16")
    m.dropout = p
    fn = get_sde_loss_fn(sch, train=True)
    res = {}
    for key, env in (("parts", {"FDIFF_TR_DX_GEMM": "0", "FDIFF_TR_ATTN_OH": "0"}), ("gemm", {"FDIFF_TR_DX_GEMM": "1"}),
                     ("gemm2", {"FDIFF_TR_DX_GEMM": "1"})):
        monkeypatch.delenv("FDIFF_TR_ATTN_OH", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m.zero_grad()
        torch.manual_seed(77)
        loss = fn(m, batch_of(X, t), noise=dev(z)).item()
        assert m.train_mode_effective == "bf16"
        res[key] = (loss, m.grads.clone(), _grads_of(m))
    assert res["gemm"][0] == res["gemm2"][0] and torch.equal(res["gemm"][1], res["gemm2"][1]), "not bit-reproducible"
    assert res["gemm"][0] == res["parts"][0]                       # the forward does not depend on the backward form
    assert not torch.equal(res["gemm"][1], res["parts"][1]) or cfg["L"] == 1, "the product form did not run (same bits as the parts)"
    worst = max(np.abs(res["gemm"][2][k] - r).max() / max(np.abs(r).max(), 1e-20) for k, r in res["parts"][2].items())
    _log(f"[parity] d x as rows . in_proj^T vs fp32 partial tensors ({tag}): worst max-rel over the tensors {worst:.3e}")
    assert worst <= 2e-3, worst


@pytest.mark.parametrize("cfg,B,p", [(dict(T=100, C=12, D=72, L=3, H=12), 9, 0.1), (dict(T=37, C=5, D=72, L=2, H=12), 7, 0.0),
                                      (dict(T=48, C=3, D=32, L=2, H=4), 5, 0.1)])
def test_ffn_f_split_agrees_with_the_unsplit_kernels(monkeypatch, cfg, B, p):
    """Below CUs / 4 blocks of 64 tokens the FFN kernels split the hidden dimension of a block over a producer / finisher pair of
    workgroups that exchange a partial accumulator through global memory (struct FSplit in fd_train_bf16.hip).  Same Philox key:
    the split step is bit-reproducible (no atomics in the sums: the finisher adds own half + partner's half in a fixed order),
    its loss equals the unsplit one to fp32 rounding and its gradients agree to the bf16 flips a different summation order of the
    FFN output causes downstream: a last-bit difference in the FFN output flips single relu / activity decisions of the next layer, so single
    elements of the linear1 gradients move by per cent (bounds: this file's per-tensor max-rel 8e-2, and l2-rel 1.5e-2; measured values logged)."""
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    tag = f"T{cfg['T']}_D{cfg['D']}_p{p}"
    X = W.randn(f"fs_x_{tag}", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn(f"fs_z_{tag}", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform(f"fs_t_{tag}", (B,), 3, 0.05, 1.0)
    m, sch, _ = make_model(cfg, precision="bf16")
    m.dropout = p
    fn = get_sde_loss_fn(sch, train=True)
    res = {}
    for mode in ("0", "1", "1"):
        monkeypatch.setenv("FDIFF_TR_FSPLIT", mode)
        m.zero_grad()
        torch.manual_seed(55)
        loss = fn(m, batch_of(X, t), noise=dev(z)).item()
        assert m.train_mode_effective == "bf16"
        g = m.grads.clone()
        if mode in res:
            assert loss == res[mode][0] and torch.equal(g, res[mode][1]), "F-split step is not bit-reproducible"
        res[mode] = (loss, g, _grads_of(m))
    assert abs(res["1"][0] - res["0"][0]) <= 2e-4 * abs(res["0"][0]), (res["1"][0], res["0"][0])
    assert not torch.equal(res["0"][1], res["1"][1]), "the split form did not run (same bits as the unsplit kernels)"
    rows = [(np.abs(res["1"][2][k] - r).max() / max(np.abs(r).max(), 1e-20),
             np.linalg.norm(res["1"][2][k] - r) / max(np.linalg.norm(r), 1e-20), k) for k, r in res["0"][2].items()]
    wm, wl = max(rows), max(rows, key=lambda x: x[1])
    _log(f"[parity] FFN F-split vs unsplit kernels ({tag}): loss {res['1'][0]:.6f} vs {res['0'][0]:.6f}, worst max-rel {wm[0]:.3e} ({wm[2]}), "
         f"worst l2-rel {wl[1]:.3e} ({wl[2]})")
    assert wm[0] <= 8e-2 and wl[1] <= 1.5e-2, (wm, wl)


def test_ffn_f_split_survives_a_cotenant_kernel_and_its_fence_form_agrees(monkeypatch):
    """The F-split finisher waits for a producer workgroup of the SAME grid (a bounded spin, struct FSplit).  Stress: the split at
    its largest allowed grid (125 token blocks -> 250 workgroups on 256 CUs, both FFN kernels: FDIFF_TR_FSPLIT=2) while a
    co-tenant stream keeps every CU busy with large GEMMs, 150 optimizer-free steps = 150 x 2 layers x 2 kernels x 500 hand-overs;
    every step must reproduce the first one bit for bit, no hand-over may time out (fd_ctx_check), and the release / acquire
    FENCE form of the hand-over (FDIFF_TR_FSPLIT_FENCE=1, the memory-model-conforming fallback) must give the same bits as the
    per-element coherent accesses (same sums in the same order)."""
    from fourierdiffusion_amd import _C
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    cfg, B = dict(T=100, C=12, D=72, L=2, H=12), 80
    X = W.randn("fsx_x", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn("fsx_z", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform("fsx_t", (B,), 3, 0.05, 1.0)
    m, sch, _ = make_model(cfg, precision="bf16")
    fn = get_sde_loss_fn(sch, train=True)
    bt = batch_of(X, t)
    zd = dev(z)
    ctx, _h = m._engine()

    def step():
        m.zero_grad()
        torch.manual_seed(91)
        loss = fn(m, bt, noise=zd)
        return loss, m.grads.clone()
    monkeypatch.setenv("FDIFF_TR_FSPLIT", "0")
    l0, g0 = step()
    monkeypatch.setenv("FDIFF_TR_FSPLIT", "2")
    l1, g1 = step()
    assert not torch.equal(g0, g1), "the split form did not run"
    assert abs(l1.item() - l0.item()) <= 2e-4 * abs(l0.item())
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    stop_after = 150
    bad = 0
    for it in range(stop_after):
        with torch.cuda.stream(side):
            for _ in range(4):
                a = torch.mm(a, a) * 1e-4          # ~140 GFLOP each: a few hundred microseconds of every CU
        l, g = step()
        bad += int(not torch.equal(g, g1)) + int(l.item() != l1.item())
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} of {stop_after} split steps beside the co-tenant kernel differ from the solo split step"
    assert _C.lib().fd_ctx_check(ctx) == 0, _C.lib().fd_last_error(ctx)
    monkeypatch.setenv("FDIFF_TR_FSPLIT_FENCE", "1")
    lf, gf = step()
    torch.cuda.synchronize()
    assert lf.item() == l1.item() and torch.equal(gf, g1), "fence form and coherent-access form of the hand-over disagree"
    assert _C.lib().fd_ctx_check(ctx) == 0


def test_ffn_f_split_timeout_is_reported_not_hung(monkeypatch):
    """A producer that never raises its flags (test hook FDIFF_TR_FSPLIT_TEST_STALL=1: what an unscheduled or faulted producer
    looks like to the finisher) must not hang the device: the finisher gives up after FDIFF_TR_FSPLIT_TIMEOUT_MS, the step
    completes, and the NEXT call on the context fails with FD_ERR_STATE naming the token block; after that the context works
    again and reproduces the healthy gradient."""
    from fourierdiffusion_amd import _C
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    cfg, B = dict(T=100, C=12, D=72, L=2, H=12), 9
    X = W.randn("fst_x", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn("fst_z", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform("fst_t", (B,), 3, 0.05, 1.0)
    m, sch, _ = make_model(cfg, precision="bf16")
    fn = get_sde_loss_fn(sch, train=True)
    ctx, _h = m._engine()
    lib = _C.lib()

    def step():
        m.zero_grad()
        torch.manual_seed(92)
        fn(m, batch_of(X, t), noise=dev(z))
        torch.cuda.synchronize()
        return m.grads.clone()
    good = step()
    assert lib.fd_ctx_check(ctx) == 0
    monkeypatch.setenv("FDIFF_TR_FSPLIT_TEST_STALL", "1")
    monkeypatch.setenv("FDIFF_TR_FSPLIT_TIMEOUT_MS", "5")
    import time
    t0 = time.perf_counter()
    stalled = step()                                   # completes: every wait is bounded
    assert time.perf_counter() - t0 < 5.0
    assert not torch.equal(stalled, good)
    monkeypatch.delenv("FDIFF_TR_FSPLIT_TEST_STALL")
    monkeypatch.delenv("FDIFF_TR_FSPLIT_TIMEOUT_MS")
    with pytest.raises(_C.FdError, match="F-split hand-over timed out at token block"):
        step()                                         # the entry check of the next training call reports it ...
    assert lib.fd_ctx_check(ctx) == 0                  # ... once
    assert torch.equal(step(), good)


def test_bf16_training_range_in_series_length():
    """The bf16 attention backward keeps a head's images and keep bits of the whole series in LDS: 592 time steps fit; beyond that
    the model trains on the exact-f32 path and says so (no silent precision switch the other way either)."""
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    for T, want in ((592, "bf16"), (600, "fp32")):
        cfg = dict(T=T, C=2, D=72, L=1, H=12)
        m, sch, _ = make_model(cfg, precision="bf16")
        X = W.randn(f"rng_x_{T}", (1, T, 2), 3)
        loss = get_sde_loss_fn(sch, train=True)(m, batch_of(X, W.uniform(f"rng_t_{T}", (1,), 3, 0.05, 1.0)))
        assert np.isfinite(loss.item()) and m.train_mode_effective == want, (T, m.train_mode_effective)
        assert float(m.grads.abs().max()) > 0 and bool(torch.isfinite(m.grads).all())


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_training_gradients_are_bit_reproducible(prec):
    """No float atomics on either training path: two identical forward+backward runs (dropout on, same Philox key) give
    bit-identical losses and gradients, and so does a run after unrelated work in between."""
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    cfg, B = dict(T=100, C=12, D=72, L=3, H=12), 16
    X = W.randn("rep_x", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn("rep_z", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform("rep_t", (B,), 3, 0.05, 1.0)
    m, sch, _ = make_model(cfg, precision=prec)
    fn = get_sde_loss_fn(sch, train=True)
    outs = []
    for rep in range(3):
        m.zero_grad()
        torch.manual_seed(31)
        loss = fn(m, batch_of(X, t), noise=dev(z))
        outs.append((loss.item(), m.grads.clone()))
        if rep == 1:
            torch.randn(1 << 20, device=DEV).sum().item()            # unrelated kernels in between
    assert outs[0][0] == outs[1][0] == outs[2][0]
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][1], outs[2][1])
    assert float(outs[0][1].abs().max()) > 0


@pytest.mark.parametrize("shape,lw", [(dict(T=100, C=12, D=72, L=2, H=12), False), (dict(T=252, C=6, D=72, L=2, H=12), True),
                                      (dict(T=64, C=28, D=72, L=1, H=12), False), (dict(T=37, C=3, D=60, L=1, H=12), False)])
def test_fused_training_call_equals_forward_loss_backward(shape, lw):
    """fd_score_train_dsm (training forward + DSM loss + backward in one call, the unembedder / loss / unembedder-backward
    as one kernel) against the three calls it replaces, same Philox key (dropout on): the loss to 1e-6 relative and the
    unembedder's own gradients to 2e-4 of their maximum (the same fp32 arithmetic in another summation order, read back through
    the 0.25 both forms accumulate onto: one ulp of 0.25 is 3e-5 of a 1e-3 gradient), every other
    gradient tensor to 1e-2 of its maximum (the head's d h differs in the last fp32 bit, which flips bf16 roundings
    downstream: measured <= 4e-3; bf16 against exact f32 is 8e-2), with a gradient weight and in accumulate mode."""
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    B = 9
    X = W.randn("fu_x", (B, shape["T"], shape["C"]), 5)
    z = W.randn("fu_z", (B, shape["T"], shape["C"]), 5)
    t = W.uniform("fu_t", (B,), 5, 0.05, 1.0)
    m, sch, _ = make_model(shape, precision="bf16")
    fn = get_sde_loss_fn(sch, train=True, likelihood_weighting=lw)
    res = {}
    for mode in ("fused", "three"):
        m._no_fused_dsm = mode == "three"
        if m.grads is None:
            m.grads = torch.zeros_like(m._flat)
        m.grads.fill_(0.25)                                         # accumulate semantics: both forms add to what is there
        torch.manual_seed(77)
        loss = fn(m, batch_of(X, t), noise=dev(z), grad_weight=1.5)
        res[mode] = (loss.item(), m.grads.clone())
    assert m.train_mode_effective == "bf16"
    m._no_fused_dsm = False
    lf, lt = res["fused"][0], res["three"][0]
    assert abs(lf - lt) <= 1e-6 * abs(lt), (lf, lt)
    worst = 0.0
    for name, off, numel, shp, _ in m._layout:
        a = res["fused"][1][off:off + numel] - 0.25
        b = res["three"][1][off:off + numel] - 0.25
        scale = float(b.abs().max())
        if scale == 0.0:
            assert float(a.abs().max()) == 0.0, name
            continue
        err = float((a - b).abs().max()) / scale
        worst = max(worst, err)
        assert err <= (2e-4 if name.startswith("unembedder") else 1e-2), (name, err)
    _log(f"[parity] fused train call vs three calls {shape} lw={lw}: loss rel {abs(lf - lt) / abs(lt):.2e}, worst tensor max-err/max {worst:.2e}")


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_zero_grad_is_lazy_and_the_next_backward_overwrites(precision):
    """ScoreModule.zero_grad() launches nothing: the next backward runs in the engine's overwrite form.  Same gradients, bit for bit, as
    an explicit fill + accumulate; a reader of ``grads`` in between sees zeros; a second backward without zero_grad() accumulates."""
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    cfg = dict(T=100, C=12, D=72, L=2, H=12)
    B = 6
    m, sch, _ = make_model(cfg, precision="bf16")
    m.train_precision = precision
    fn = get_sde_loss_fn(sch, train=True)
    X = dev(W.randn("lz_x", (B, cfg["T"], cfg["C"]), 11))
    t = dev(W.uniform("lz_t", (B,), 11, 0.05, 1.0))
    z = dev(W.randn("lz_z", (B, cfg["T"], cfg["C"]), 11))

    def run(lazy):
        torch.manual_seed(77)
        if lazy:
            m.zero_grad()
            assert m._zero_pending or m._grads is None
        else:
            if m.grads is None:
                m.grads = torch.zeros_like(m.flat_parameters)
            m.grads.fill_(0.0)
            assert not m._zero_pending
        loss = fn(m, DiffusableBatch(X=X, timesteps=t), noise=z)
        return float(loss), m.grads.clone()
    l0, g0 = run(False)            # (also allocates the buffer)
    m.grads.fill_(123.0)           # garbage the overwrite form must not see
    l1, g1 = run(True)
    assert l0 == l1 and torch.equal(g0, g1) and float(g1.abs().max()) > 0
    m.zero_grad()
    assert float(m.grads.abs().max()) == 0.0 and not m._zero_pending       # a reader in between gets the zeros
    torch.manual_seed(77)
    fn(m, DiffusableBatch(X=X, timesteps=t), noise=z)
    torch.manual_seed(77)
    fn(m, DiffusableBatch(X=X, timesteps=t), noise=z)                        # no zero_grad(): accumulates
    assert torch.allclose(m.grads, 2 * g0, rtol=1e-6, atol=0)


def test_fused_training_call_overwrite_mode_through_the_c_abi():
    """fd_score_train_dsm with accumulate = 0 (a C-ABI caller that does not zero its gradient buffer): bit-identical to
    accumulate = 1 on a zeroed buffer, whatever the buffer held before; same Philox key."""
    import ctypes as C

    from fourierdiffusion_amd import _C
    cfg = dict(T=100, C=12, D=72, L=2, H=12)
    B = 6
    m, sch, _ = make_model(cfg, precision="bf16")
    m.train()
    ctx, h = m._engine()
    assert m.train_mode_effective == "bf16"
    lib = _C.lib()
    X = dev(W.randn("ow_x", (B, cfg["T"], cfg["C"]), 9))
    t = dev(W.uniform("ow_t", (B,), 9, 0.05, 1.0))
    z = dev(W.randn("ow_z", (B, cfg["T"], cfg["C"]), 9))
    xn, target, std = sch.perturb(X, t, noise=z)
    out = []
    for acc, fill in ((1, 0.0), (0, 123.0)):
        grads = torch.full_like(m.flat_parameters, fill)
        loss = torch.empty(1, device=DEV)
        rc = lib.fd_score_train_dsm(h, xn.data_ptr(), t.data_ptr(), target.data_ptr(), std.data_ptr(), 0, C.c_float(1.0), B,
                                    C.c_float(0.1), 4242, 17, loss.data_ptr(), grads.data_ptr(), acc,
                                    torch.cuda.current_stream(DEV).cuda_stream)
        _C.check(rc, ctx)
        out.append((loss.item(), grads.clone()))
    assert out[0][0] == out[1][0]
    frozen = [(off, numel) for name, off, numel, shp, _ in m._layout if name.startswith("time_encoder.W")]
    g1 = out[1][1].clone()
    for off, numel in frozen:                       # (requires_grad = False: its slot is cleared, not computed)
        assert float(g1[off:off + numel].abs().max()) == 0.0
    assert torch.equal(out[0][1], g1)
    assert float(g1.abs().max()) > 0


def test_dropout_forward_backward_consistency_bf16():
    """dropout p=0.1 in the bf16 path: the stored keep bits are what the backward uses.  grad . v against a central
    difference of the (bf16) loss along a random direction, same Philox key: 10 % tolerance (bf16 forward noise)."""
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    cfg = dict(T=40, C=4, D=72, L=2, H=12)
    m, sch, _ = make_model(cfg, precision="bf16")
    assert m.dropout == pytest.approx(0.1)
    B = 32
    X = W.randn("drb_x", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn("drb_z", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform("drb_t", (B,), 3, 0.2, 1.0)
    fn = get_sde_loss_fn(sch, train=True)

    def loss_at(seed, backward):
        torch.manual_seed(seed)
        return fn(m, batch_of(X, t), noise=dev(z), backward=backward).item()

    m.zero_grad()
    l0 = loss_at(11, True)
    grads = m.grads.clone()
    assert m.train_mode_effective == "bf16"
    # (the training call with backward runs the fused loss head, the one without it the stand-alone loss kernel: the same
    # numbers summed in another order; each form is bit-reproducible)
    l0f = loss_at(11, False)
    assert l0f == loss_at(11, False) and abs(l0f - l0) <= 1e-6 * abs(l0)
    assert abs(loss_at(12, False) - l0) > 1e-5 * abs(l0)
    m.dropout = 0.0
    l_nodrop = loss_at(11, False)
    m.dropout = 0.1
    assert l_nodrop != l0
    # direction of steepest ascent: the loss change eps * |g| must stand clear of the bf16 rounding noise of the two forward
    # evaluations (~1e-3 of the loss), which a random direction in 3e5 dimensions does not
    v = grads / grads.norm()
    eps = 0.1 / max(float(grads.norm()), 1e-6) * abs(l0)         # ~10 % loss change per side
    base = m.flat_parameters.clone()
    m.flat_parameters.copy_(base + eps * v); m.mark_parameters_changed()
    lp = loss_at(11, False)
    m.flat_parameters.copy_(base - eps * v); m.mark_parameters_changed()
    lm = loss_at(11, False)
    m.flat_parameters.copy_(base); m.mark_parameters_changed()
    fd = (lp - lm) / (2 * eps)
    an = float((grads * v).sum())
    _log(f"[parity] bf16 dropout consistency: central difference {fd:.5e} vs grad.v {an:.5e}")
    assert abs(fd - an) <= 0.1 * max(abs(fd), abs(an)) + 1e-4, (fd, an)


def test_dropout_on_gradients_per_tensor_bf16():
    """Dropout ON (p = 0.1), bf16 path, per parameter tensor: the directional derivative of the (same-key) loss along the
    tensor's own gradient block u_k = g_k / |g_k| must equal |g_k| -- a wrong keep-bit orientation, a missing keep scale or a
    gradient that ignores the masks in ONE tensor (the weight-gradient kernel and the two input-gradient kernels read the
    bits in three different layouts) changes that tensor's projection while the whole-gradient check above can still pass.
    Central differences with a 10 % loss change per side; 15 % tolerance (bf16 forward noise + curvature)."""
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    cfg = dict(T=40, C=4, D=72, L=2, H=12)
    m, sch, _ = make_model(cfg, precision="bf16")
    B = 32
    X = W.randn("drt_x", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn("drt_z", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform("drt_t", (B,), 3, 0.2, 1.0)
    fn = get_sde_loss_fn(sch, train=True)

    def loss_at(backward):
        torch.manual_seed(21)
        return fn(m, batch_of(X, t), noise=dev(z), backward=backward).item()

    m.zero_grad()
    l0 = loss_at(True)
    assert m.train_mode_effective == "bf16" and m.dropout == pytest.approx(0.1)
    grads = m.grads.clone()
    base = m.flat_parameters.clone()
    views = {name: (off, numel) for name, off, numel, shape, _ in m._layout}
    keys = ["backbone.layers.0.linear1.weight", "backbone.layers.1.linear2.weight", "backbone.layers.0.self_attn.in_proj_weight",
            "backbone.layers.1.self_attn.out_proj.weight", "backbone.layers.0.norm1.weight", "backbone.layers.1.linear1.bias",
            "embedder.weight", "pos_encoder.embedding.weight", "unembedder.weight"]
    worst = 0.0
    for k in keys:
        off, numel = views[k]
        gk = grads[off:off + numel]
        nk = float(gk.norm())
        assert nk > 0, k
        eps = 0.1 * abs(l0) / nk
        v = torch.zeros_like(base)
        v[off:off + numel] = gk / nk
        m.flat_parameters.copy_(base + eps * v); m.mark_parameters_changed()
        lp = loss_at(False)
        m.flat_parameters.copy_(base - eps * v); m.mark_parameters_changed()
        lm = loss_at(False)
        fd = (lp - lm) / (2 * eps)
        rel = abs(fd - nk) / nk
        worst = max(worst, rel)
        _log(f"[parity] bf16 dropout-on gradient of {k}: central difference {fd:.4e} vs |g_k| {nk:.4e} (rel {rel:.3e})")
        assert rel <= 0.15, (k, fd, nk)
    m.flat_parameters.copy_(base); m.mark_parameters_changed()
    _log(f"[parity] bf16 dropout-on per-tensor directional derivatives: worst relative deviation {worst:.3e}")


def test_short_optimisation_run_bf16_tracks_f32():
    """40 AdamW steps on a fixed synthetic batch with the bf16 and the exact-f32 training kernels (dropout 0, same t, z per
    step): both losses fall, and the bf16 loss curve stays within 5 % of the f32 one."""
    from fourierdiffusion_amd.optim import FusedAdamW
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    cfg, B = dict(T=48, C=4, D=72, L=2, H=12), 64
    X = W.randn("opt_x", (B, cfg["T"], cfg["C"]), 3)
    curves = {}
    for prec in ("fp32", "bf16"):
        torch.manual_seed(0)
        m, sch, _ = make_model(cfg, precision=prec)
        m.dropout = 0.0
        opt = FusedAdamW(m, lr=2e-3, max_grad_norm=1.0)
        fn = get_sde_loss_fn(sch, train=True)
        g = torch.Generator(device="cpu").manual_seed(9)
        losses = []
        for step in range(40):
            t = torch.rand(B, generator=g) * 0.9 + 0.05
            z = torch.randn(B, cfg["T"], cfg["C"], generator=g)
            m.zero_grad()
            losses.append(fn(m, batch_of(X, t), noise=z.to(DEV)).item())
            opt.step()
        curves[prec] = np.array(losses)
    for prec, c in curves.items():
        assert c[-5:].mean() < 0.8 * c[:5].mean(), (prec, c[:5], c[-5:])
    rel = np.abs(curves["bf16"] - curves["fp32"]) / curves["fp32"]
    _log(f"[parity] bf16 vs f32 loss curve over 40 AdamW steps: max rel diff {rel.max():.3e}, final {curves['bf16'][-1]:.4f} vs {curves['fp32'][-1]:.4f}")
    assert rel.max() <= 5e-2, rel.max()


@pytest.mark.parametrize("name,B", [("default", 4), ("nasdaq", 2)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_default_model_gradients_vs_reference_fixture(golden, name, B, prec):
    """The one model the hydra configs name (d_model 72, 10 layers, ff 2048) against the REFERENCE's autograd gradients
    (tests/golden/grad_default.npz: per-tensor norms and every 97th element; oracle/make_golden.py gen_grad_default), at the
    ecg shape and the BASELINE training shape (nasdaq T=252, C=6).  Exact-f32 engine: 1e-3 of each tensor's max (measured 3e-5 at T=100, 6e-4 at T=252: 10 layers
    of fp32 reordering); bf16 engine: the bf16 tolerances of this file."""
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    from oracle.make_golden import GRAD_STRIDE
    g = golden("grad_default")
    cfg = CFG_DEFAULT if name == "default" else dict(T=252, C=6, D=72, L=10, H=12)
    X = W.randn(f"gd_x_{name}", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn(f"gd_z_{name}", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform(f"gd_t_{name}", (B,), 3, 0.05, 1.0)
    m, sch, _ = make_model(cfg, precision=prec)
    m.dropout = 0.0
    m.zero_grad()
    loss = get_sde_loss_fn(sch, train=True)(m, batch_of(X, t), noise=dev(z)).item()
    assert m.train_mode_effective == prec
    ref_loss = float(g[f"loss_{name}"])
    assert abs(loss - ref_loss) <= (2e-5 if prec == "fp32" else 1e-2) * abs(ref_loss), (loss, ref_loss)
    got = _grads_of(m)
    worst = (0.0, "")
    for key in g.files:
        if not key.startswith(f"sub_{name}/"):
            continue
        k = key.split("/", 1)[1]
        sub = g[key].astype(np.float64)
        nrm, mx = g[f"norm_{name}/{k}"]
        mine = got[k].ravel()
        e_sub = np.abs(mine[::GRAD_STRIDE] - sub).max() / max(mx, 1e-20)
        e_nrm = abs(np.linalg.norm(mine) - nrm) / max(nrm, 1e-20)
        worst = max(worst, (e_sub, k))
        if prec == "fp32":
            assert e_sub <= 1e-3 and e_nrm <= 5e-4, (k, e_sub, e_nrm)
        else:
            assert e_sub <= 8e-2 and e_nrm <= 3e-2, (k, e_sub, e_nrm)
    _log(f"[parity] {prec} gradients vs reference fixture ({name}, d_model 72, L 10): worst subset max-rel {worst[0]:.3e} ({worst[1]})")


def test_interleaved_models_and_eval_share_the_context_arena():
    """the dropout-decision buffers live in the context's workspace arena and are written by side-stream kernels.
    Training steps of two models with different layer counts interleaved with evaluation forwards (which carve the same
    arena) must give the same gradients as each model trained alone with the same keys -- a mask kernel that started before
    the other call's kernels had finished with the arena, or a wait on the wrong event, would change them."""
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    cfgs = [dict(T=40, C=4, D=72, L=2, H=12), dict(T=56, C=3, D=72, L=3, H=12)]
    data = []
    for i, cfg in enumerate(cfgs):
        X = W.randn(f"ilv_x{i}", (6, cfg["T"], cfg["C"]), 3)
        z = W.randn(f"ilv_z{i}", (6, cfg["T"], cfg["C"]), 3)
        t = W.uniform(f"ilv_t{i}", (6,), 3, 0.2, 1.0)
        data.append((X, z, t))

    def step(m, sch, i, seed):
        fn = get_sde_loss_fn(sch, train=True)
        X, z, t = data[i]
        m.zero_grad()
        torch.manual_seed(seed)
        loss = fn(m, batch_of(X, t), noise=dev(z), backward=True).item()
        return loss, m.grads.clone()

    alone = []
    for i, cfg in enumerate(cfgs):
        m, sch, _ = make_model(cfg, precision="bf16")
        alone.append([step(m, sch, i, 31 + k) for k in range(2)])
    models = [make_model(cfg, precision="bf16") for cfg in cfgs]
    mixed = [[], []]
    for k in range(2):
        for i in (0, 1):
            m, sch, _ = models[i]
            mixed[i].append(step(m, sch, i, 31 + k))
            other = models[1 - i][0]
            Xo, _, to = data[1 - i]
            other.eval()
            _ = other(DiffusableBatch(X=dev(Xo), timesteps=dev(to)))        # carves the arena between the training steps
    for i in (0, 1):
        for k in range(2):
            assert mixed[i][k][0] == alone[i][k][0], (i, k)
            assert torch.equal(mixed[i][k][1], alone[i][k][1]), (i, k)
