"""GPU parity: training path -- loss, parameter gradients (fd_score_backward), fused AdamW + clip.
Gradients are compared with the reference's autograd gradients (dropout forced to 0, injected t and z;
tests/golden/loss.npz) at rtol 2e-4 of each tensor's max |grad| (fp32 accumulation order differs)."""
import os

import numpy as np
import pytest
import torch

from oracle import fdiff_oracle as O
from oracle import weights as W
from oracle.make_golden import CFG_ODD, CFG_TINY, SDE_CASES

from .gpu_util import DEV, dev, host, make_model

pytestmark = pytest.mark.gpu
CFGS = {"tiny": CFG_TINY, "odd": CFG_ODD}


def batch_of(X, t):
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    return DiffusableBatch(X=dev(X), y=None, timesteps=dev(t))


@pytest.mark.parametrize("name,B", [("tiny", 5), ("odd", 3)])
def test_loss_and_gradients_vs_reference(golden, name, B):
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    g = golden("loss")
    cfg = CFGS[name]
    X = W.randn(f"loss_x_{name}", (B, cfg["T"], cfg["C"]), 3)
    z = W.randn(f"loss_z_{name}", (B, cfg["T"], cfg["C"]), 3)
    t = W.uniform(f"loss_t_{name}", (B,), 3, 0.05, 1.0)
    for ci, (kind, p) in enumerate(SDE_CASES[:2]):
        for lw in (False, True):
            tag = f"{name}_{kind}{ci}_{int(lw)}"
            m, sch, _ = make_model(cfg, kind, p, precision="fp32")
            m.dropout = 0.0
            ev = get_sde_loss_fn(sch, train=False, likelihood_weighting=lw)(m, batch_of(X, t), noise=dev(z))
            np.testing.assert_allclose(ev.item(), g[f"loss_{tag}"], rtol=2e-5)
            m.zero_grad()
            tr = get_sde_loss_fn(sch, train=True, likelihood_weighting=lw)(m, batch_of(X, t), noise=dev(z))
            np.testing.assert_allclose(tr.item(), g[f"loss_train_{tag}"], rtol=2e-5)
            if lw:
                continue
            gv = m.grad_views()
            checked = 0
            for k, gt in gv.items():
                key = f"grad_{tag}/{k}"
                if key not in g.files:
                    assert k == "time_encoder.W"                # requires_grad=False in the reference
                    assert float(gt.abs().max()) == 0.0
                    continue
                ref = g[key]
                scale = max(np.abs(ref).max(), 1e-12)
                err = np.abs(host(gt) - ref).max() / scale
                assert err < 2e-4, (tag, k, err)
                checked += 1
            assert checked == len(g.files and [f for f in g.files if f.startswith(f"grad_{tag}/")])


def test_gradient_accumulation_and_zero_grad():
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    cfg = CFG_TINY
    m, sch, _ = make_model(cfg, precision="fp32")
    m.dropout = 0.0
    X = W.randn("acc_x", (4, cfg["T"], cfg["C"]), 3)
    z = W.randn("acc_z", (4, cfg["T"], cfg["C"]), 3)
    t = W.uniform("acc_t", (4,), 3, 0.05, 1.0)
    fn = get_sde_loss_fn(sch, train=True)
    fn(m, batch_of(X, t), noise=dev(z))
    g1 = m.grads.clone()
    fn(m, batch_of(X, t), noise=dev(z))
    assert torch.allclose(m.grads, 2 * g1, rtol=1e-5, atol=1e-7)
    m.zero_grad()
    assert float(m.grads.abs().max()) == 0.0


CFG_D10 = dict(T=12, C=2, D=10, L=1, H=2)      # d_model % 4 != 0: the dropout sites cannot ride in the GEMM epilogues


@pytest.mark.parametrize("cfg", [CFG_TINY, CFG_D10], ids=["tiny", "d_model_10"])
def test_dropout_forward_backward_consistency(cfg):
    """dropout p=0.1: masks are regenerated in backward from the same Philox key.  With the key pinned through
    torch.manual_seed the loss is a deterministic function of the parameters; check grad . v against a central
    difference along a random direction v (fp32: 3 % tolerance), and that masks really are applied."""
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    m, sch, _ = make_model(cfg, precision="fp32")
    assert m.dropout == pytest.approx(0.1)
    X = W.randn("dr_x", (6, cfg["T"], cfg["C"]), 3)
    z = W.randn("dr_z", (6, cfg["T"], cfg["C"]), 3)
    t = W.uniform("dr_t", (6,), 3, 0.2, 1.0)
    fn = get_sde_loss_fn(sch, train=True)

    def loss_at(seed, backward):
        torch.manual_seed(seed)
        return fn(m, batch_of(X, t), noise=dev(z), backward=backward).item()

    m.zero_grad()
    l0 = loss_at(11, True)
    grads = m.grads.clone()
    assert loss_at(11, False) == l0                       # same key -> same masks -> same loss
    assert loss_at(12, False) != l0                       # another key -> other masks
    m.eval()
    torch.manual_seed(5)
    v = torch.randn_like(m.flat_parameters)
    for name, off, numel, _, tr in m._layout:
        if not tr:
            v[off:off + numel] = 0
    v /= v.norm()
    eps = 2e-2
    base = m.flat_parameters.clone()
    m.flat_parameters.copy_(base + eps * v); m.mark_parameters_changed()
    lp = loss_at(11, False)
    m.flat_parameters.copy_(base - eps * v); m.mark_parameters_changed()
    lm = loss_at(11, False)
    m.flat_parameters.copy_(base); m.mark_parameters_changed()
    fd = (lp - lm) / (2 * eps)
    an = float((grads * v).sum())
    assert abs(fd - an) <= 3e-2 * max(abs(fd), abs(an)) + 1e-4, (fd, an)


def test_fused_adamw_and_clip_vs_torch(golden):
    """fd_grad_sqnorm + fd_adamw_step vs torch.optim.AdamW + clip_grad_norm_ (tests/golden/optim.npz)."""
    import ctypes as C
    from fourierdiffusion_amd import _C
    g = golden("optim")
    p = dev(W.randn("opt_p", (257,), 6))
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    sq = torch.zeros(1, device=DEV)
    h = _C.ctx(p.device)
    L = _C.lib()
    for it in range(3):
        grad = dev(W.randn(f"opt_g{it}", (257,), 6) * np.float32(3.0))
        _C.check(L.fd_grad_sqnorm(h, grad.data_ptr(), 257, sq.data_ptr(), None), h)
        np.testing.assert_allclose(np.sqrt(sq.item()), g[f"total_norm_{it}"], rtol=1e-5)
        _C.check(L.fd_adamw_step(h, p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), 257, it + 1,
                                 1e-3 * (it + 1) / 3, 0.9, 0.999, 1e-8, 1e-2, sq.data_ptr(), 1.0, 1.0, 0, 0, None), h)
        np.testing.assert_allclose(host(p), g[f"param_{it}"], rtol=2e-6, atol=1e-7)
    # frozen range is left untouched
    q = p.clone()
    _C.check(L.fd_adamw_step(h, p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), 257, 4, 1e-3, 0.9, 0.999,
                             1e-8, 1e-2, None, 0.0, 1.0, 10, 20, None), h)
    assert torch.equal(p[10:20], q[10:20]) and not torch.equal(p[:10], q[:10])


@pytest.mark.parametrize("kind", ["vp", "ve"])
def test_training_changes_every_trainable_parameter_and_learns(kind):
    """tests/test_score_models.py:76-89 / test_schedulers.py:76-117 of the reference: after training every
    parameter except time_encoder.W has changed; additionally the loss on a fixed batch goes down."""
    from fourierdiffusion_amd.optim import FusedAdamW
    from fourierdiffusion_amd.sampling.sampler import DiffusionSampler
    from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
    cfg = CFG_TINY
    p = (0.1, 20.0) if kind == "vp" else (0.01, 2.0)
    m, sch, _ = make_model(cfg, kind, p, precision="fp32")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    X = W.randn("tr_x", (50, cfg["T"], cfg["C"]), 3)
    z = W.randn("tr_z", (50, cfg["T"], cfg["C"]), 3)
    t = W.uniform("tr_t", (50,), 3, 0.05, 1.0)
    opt = FusedAdamW(m, lr=1e-3, max_grad_norm=1.0)
    fn_tr = m.training_loss_fn
    fn_ev = m.validation_loss_fn
    l_before = fn_ev(m, batch_of(X, t), noise=dev(z)).item()
    for _ in range(30):
        opt.zero_grad()
        fn_tr(m, batch_of(X, t), noise=dev(z))
        opt.step()
    l_after = fn_ev(m, batch_of(X, t), noise=dev(z)).item()
    assert l_after < l_before
    after = m.state_dict()
    for k in before:
        if k == "time_encoder.W":
            assert torch.equal(before[k], after[k])
        elif k == "pos_encoder.embedding.weight":
            assert not torch.equal(before[k], after[k])
        else:
            assert not torch.equal(before[k], after[k]), k
    sampler = DiffusionSampler(score_model=m, sample_batch_size=12)
    s = sampler.sample(num_samples=48, num_diffusion_steps=10)
    assert s.shape == (48, cfg["T"], cfg["C"]) and torch.isfinite(s).all()


def test_rccl_allreduce_single_rank():
    """fd_comm_* / fd_allreduce_grads on a one-rank communicator: exercises the lazy RCCL binding (dlopen, symbol table,
    unique id, communicator init, all-reduce + scale on the caller's stream) on the hardware that is available to the
    test run; the world_size-2 averaging logic is covered on CPU (tests/test_distributed_cpu.py) and the 8-GPU run is the
    driver's."""
    import ctypes as C
    from fourierdiffusion_amd import _C
    dev = torch.device(DEV, torch.cuda.current_device()) if isinstance(DEV, str) else DEV
    ctx = _C.ctx(torch.device("cuda", torch.cuda.current_device()))
    lib = _C.lib()
    uid = (C.c_ubyte * _C.FD_COMM_ID_BYTES)()
    assert lib.fd_comm_unique_id(uid) == 0
    _C.check(lib.fd_comm_init(ctx, 0, 1, uid), ctx)
    try:
        g = torch.randn(3_200_000, device="cuda")
        ref = g.clone()
        _C.check(lib.fd_allreduce_grads(ctx, g.data_ptr(), g.numel(), 0.5, torch.cuda.current_stream().cuda_stream), ctx)
        torch.cuda.synchronize()
        assert torch.allclose(g, ref * 0.5, rtol=0, atol=0)
    finally:
        _C.check(lib.fd_comm_destroy(ctx), ctx)


def test_fused_dropout_masks_equal_the_standalone_kernel(tmp_path):
    """The training forward applies its dropout sites in the GEMM / split-K epilogues; FDIFF_GEMM=valu selects the VALU GEMM,
    which cannot fuse, so the same sites run as stand-alone fd_k_dropout launches.  Same Philox key -> the two outputs must
    agree to fp32 GEMM rounding (a single differing mask bit would show as an O(1) difference)."""
    import subprocess
    import sys
    script = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
from tests.gpu_util import make_model, dev
from oracle import weights as W
from oracle.make_golden import CFG_ODD
from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
m, sch, _ = make_model(CFG_ODD, precision="fp32")
m.train()
X = W.randn("fd_x", (5, CFG_ODD["T"], CFG_ODD["C"]), 3); t = W.uniform("fd_t", (5,), 3, 0.2, 1.0)
torch.manual_seed(77)
out = m(DiffusableBatch(X=dev(X), y=None, timesteps=dev(t)))
np.save(sys.argv[1], out.detach().cpu().numpy())
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for name, env in (("fused", {}), ("valu", {"FDIFF_GEMM": "valu"})):
        path = str(tmp_path / f"{name}.npy")
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", script, path], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
    assert np.abs(outs[0]).max() > 0.1
    np.testing.assert_allclose(outs[0], outs[1], atol=2e-5, rtol=0)


def _dp_worker(rank, world, port, out_dir, root):
    import sys
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), FDIFF_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from fourierdiffusion_amd.dataloaders.datamodules import TensorDatamodule
    from fourierdiffusion_amd.trainer import Trainer
    from tests.gpu_util import make_model
    m, _, _ = make_model(CFG_TINY, precision="fp32")
    X = torch.from_numpy(W.randn("dp_x", (48, CFG_TINY["T"], CFG_TINY["C"]), 5))
    dm = TensorDatamodule(X_train=X, batch_size=16, fourier_transform=True, standardize=True)
    torch.manual_seed(7)                                   # same shuffling / Philox key on every rank
    before = m.flat_parameters.clone()
    tr = Trainer(max_epochs=2, gradient_clip_val=1.0, grad_exchange="torch", enable_progress_bar=False)
    tr.fit(m, dm)
    assert dist.get_backend() == "gloo" and tr.dist.world == world and tr.global_step == 6
    seen = torch.cat([b.X[:, 0, 0].cpu() for b in dm.train_dataloader()])
    torch.save({"params": m.flat_parameters.cpu(), "moved": float((m.flat_parameters - before).abs().max()),
                "loss": tr.logged["train/loss"], "seen": seen}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_training_on_one_gpu(tmp_path):
    """Data-parallel training end to end with two ranks sharing cuda:0 (rendezvous and gradient exchange over gloo,
    FDIFF_DIST_BACKEND; on the 8-GPU node the exchange is fd_allreduce_grads over RCCL): every rank trains on its own shard
    of each batch with its own Philox counter range, the flat gradient is averaged, and the fused AdamW leaves BIT-IDENTICAL
    parameters on both ranks after 6 optimizer steps."""
    import socket

    import torch.multiprocessing as mp
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path), root), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in (0, 1))
    assert torch.equal(r0["params"], r1["params"])
    assert r0["moved"] > 1e-4 and r0["loss"] == r1["loss"] and np.isfinite(r0["loss"])
    assert r0["seen"].numel() + r1["seen"].numel() == 48 and not set(r0["seen"].tolist()) & set(r1["seen"].tolist())
