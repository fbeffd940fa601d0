"""CPU, world_size 2 over gloo: the data-parallel host logic (batch sharding, flat-gradient averaging, scalar
averaging, unique-id broadcast plumbing, sample-batch shard ranges).  The RCCL transport itself (fd_allreduce_grads)
needs GPUs; what is checked here is everything around it, through the same GradExchange interface."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from fourierdiffusion_amd.dataloaders.datamodules import BatchLoader, DiffusionDataset
    from fourierdiffusion_amd.parallel import GradExchange, init_process_group, shard_range
    e = init_process_group(backend="gloo")
    assert (e.rank, e.world) == (rank, world) and dist.get_backend() == "gloo"
    ex = GradExchange(e, backend="torch")
    # flat gradient buffer: rank r holds (r+1) * base -> mean = 1.5 * base for world 2
    base = torch.arange(1000, dtype=torch.float32) / 7.0
    g = base * (rank + 1)
    ex.all_reduce_mean(g)
    assert torch.allclose(g, base * (sum(range(1, world + 1)) / world), rtol=1e-6)
    assert ex.all_reduce_scalar_mean(float(rank)) == pytest.approx((world - 1) / 2)
    # unique-id style broadcast (host bytes from rank 0)
    uid = torch.arange(128, dtype=torch.uint8) if rank == 0 else torch.zeros(128, dtype=torch.uint8)
    dist.broadcast(uid, src=0)
    assert uid.tolist() == list(range(128))
    # the same seed on every rank gives the same permutation, ranks take disjoint strided slices of each batch
    torch.manual_seed(123)
    ds = DiffusionDataset(torch.arange(37 * 4 * 2, dtype=torch.float32).reshape(37, 4, 2))
    seen = torch.cat([b.X[:, 0, 0] for b in BatchLoader(ds, 8, shuffle=True, rank=rank, world=world)])
    torch.save(seen, os.path.join(out_dir, f"seen_{rank}.pt"))
    lo, hi = shard_range(7, rank, world)
    torch.save(torch.tensor([lo, hi]), os.path.join(out_dir, f"range_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_data_parallel_logic(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    seen = [torch.load(tmp_path / f"seen_{r}.pt") for r in range(world)]
    allv = torch.cat(seen)
    assert allv.numel() == 37 and torch.equal(allv.sort().values, torch.arange(37, dtype=torch.float32) * 8)
    assert abs(seen[0].numel() - seen[1].numel()) <= 5          # per-batch strided split of 5 batches
    ranges = [torch.load(tmp_path / f"range_{r}.pt").tolist() for r in range(world)]
    assert ranges == [[0, 4], [4, 7]]


def test_shard_range_covers_everything():
    from fourierdiffusion_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 50):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_rank_offsets_are_disjoint():
    """Ranks share the seed (so the Philox key is the same) and must use disjoint counter ranges."""
    from fourierdiffusion_amd import _rng
    torch.manual_seed(9)
    _rng.set_rank(0)
    k0, o0 = _rng.stream()
    torch.manual_seed(9)
    _rng.set_rank(3)
    k3, o3 = _rng.stream()
    _rng.set_rank(0)
    assert k0 == k3 and o0 == 0 and o3 == 3 << 56


# ---------------------------------------------------------------------------------------------------------------------
# Trainer.fit under world_size 2 with a LAST BATCH SMALLER THAN THE WORLD (n % batch_size == 1, the reference's ECG case:
# 87 553 rows, batch 64): every rank must join every all-reduce (an empty slice contributes zero), slices of unequal size
# are weighted by n_local / n_global, and a partial accumulation window is flushed at the end of the epoch.
class _QuadModel:
    """Stand-in for ScoreModule on the CPU: loss = mean_i |w - x_i|^2 / 2 over the batch; the gradient of a batch is
    w - mean(x).  Implements exactly the surface Trainer.fit touches."""

    def __init__(self, dim):
        self.flat_parameters = torch.zeros(dim, dtype=torch.float64)
        self.grads = None
        self.device = torch.device("cpu")
        self.steps = []

    def to(self, dev):
        return self

    def eval(self):
        return self

    def zero_grad(self):
        if self.grads is not None:
            self.grads.zero_()

    def training_step(self, batch, bi, grad_weight=1.0):
        x = batch.X.reshape(len(batch), -1).double()
        if self.grads is None:
            self.grads = torch.zeros_like(self.flat_parameters)
        self.grads += grad_weight * (self.flat_parameters - x.mean(0))
        return 0.5 * ((self.flat_parameters - x) ** 2).sum(1).mean()

    def validation_step(self, batch, bi):
        return torch.zeros(())

    def configure_optimizers(self):
        model = self

        class _SGD:
            lr = base_lr = 0.5
            max_grad_norm = None

            def step(self, grad_scale=1.0):
                model.steps.append(model.grads.clone() * grad_scale)
                model.flat_parameters -= self.lr * grad_scale * model.grads

            def state_dict(self):
                return {}
        return {"optimizer": _SGD(), "lr_scheduler": {"scheduler": lambda step: 1.0, "interval": "step"}}


def _fit_worker(rank, world, port, out_dir, n, bs, accum):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), FDIFF_DIST_BACKEND="gloo", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    import torch.distributed as dist
    from fourierdiffusion_amd.dataloaders.datamodules import TensorDatamodule
    from fourierdiffusion_amd.trainer import Trainer
    torch.manual_seed(5)
    X = torch.randn(n, 3, 2)
    dm = TensorDatamodule(X_train=X, X_test=X[:4], batch_size=bs)
    model = _QuadModel(6)
    tr = Trainer(max_epochs=2, grad_exchange="torch", accumulate_grad_batches=accum, enable_progress_bar=False)
    torch.manual_seed(77)                    # the shuffle permutation: same on every rank
    tr.fit(model, dm)
    torch.save({"w": model.flat_parameters, "steps": tr.global_step, "loss": tr.logged["train/loss"]},
               os.path.join(out_dir, f"fit_{rank}.pt"))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("n,bs,accum", [(17, 8, 1), (9, 4, 2), (21, 5, 1)])
def test_world2_fit_small_last_batch_matches_single_process(tmp_path, n, bs, accum):
    """(17, 8): last global batch has ONE sample -> rank 1's slice is empty.  (9, 4, accum 2): 3 batches, the last
    accumulation window is partial.  (21, 5): uneven 3 + 2 slices in every batch.  Both ranks must finish (no hang), end
    with identical parameters, and match the single-process run on the full batches to fp64 rounding."""
    world = 2
    mp.spawn(_fit_worker, args=(world, _free_port(), str(tmp_path), n, bs, accum), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"fit_{r}.pt") for r in range(world)]
    single = tmp_path / "single"
    single.mkdir()
    _fit_single(str(single), n, bs, accum)
    ref = torch.load(single / "fit_0.pt")
    assert res[0]["steps"] == res[1]["steps"] == ref["steps"]
    assert torch.equal(res[0]["w"], res[1]["w"])
    assert torch.allclose(res[0]["w"], ref["w"], rtol=0, atol=1e-12)
    assert res[0]["loss"] == pytest.approx(ref["loss"], rel=1e-9)


def _fit_single(out_dir, n, bs, accum):
    saved = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    for k in saved:
        os.environ.pop(k, None)
    try:
        from fourierdiffusion_amd.dataloaders.datamodules import TensorDatamodule
        from fourierdiffusion_amd.trainer import Trainer
        torch.manual_seed(5)
        X = torch.randn(n, 3, 2)
        dm = TensorDatamodule(X_train=X, X_test=X[:4], batch_size=bs)
        model = _QuadModel(6)
        tr = Trainer(max_epochs=2, grad_exchange="torch", accumulate_grad_batches=accum, enable_progress_bar=False)
        torch.manual_seed(77)
        tr.fit(model, dm)
        torch.save({"w": model.flat_parameters, "steps": tr.global_step, "loss": tr.logged["train/loss"]},
                   os.path.join(out_dir, "fit_0.pt"))
    finally:
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v


def test_batchloader_every_rank_yields_every_global_batch():
    from fourierdiffusion_amd.dataloaders.datamodules import BatchLoader, DiffusionDataset
    ds = DiffusionDataset(torch.arange(65 * 2, dtype=torch.float32).reshape(65, 2, 1))
    counts = []
    for rank in range(4):
        torch.manual_seed(0)
        items = list(BatchLoader(ds, 64, shuffle=True, rank=rank, world=4))
        counts.append([len(b) for b in items])
        assert [b.global_size for b in items] == [64, 1]
    assert counts == [[16, 1], [16, 0], [16, 0], [16, 0]]


# ---------------------------------------------------------------------------------------------------------------------
# a rank whose slice of a small last batch is empty runs no training step and so draws no Philox keys
# from torch's global generator; the shuffle of the next epoch must not depend on that, and the generators must stay in step.
class _KeyDrawingModel(_QuadModel):
    """Like ScoreModule.training_step: one key for the perturbation noise, one for the dropout masks."""

    def __init__(self, dim):
        super().__init__(dim)
        self.seen = []

    def training_step(self, batch, bi, grad_weight=1.0):
        from fourierdiffusion_amd import _rng
        _rng.stream()
        _rng.stream()
        self.seen.append(batch.X[:, 0, 0].clone())
        return super().training_step(batch, bi, grad_weight)


def _rng_fit_worker(rank, world, port, out_dir, n, bs, epochs):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), FDIFF_DIST_BACKEND="gloo", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    import torch.distributed as dist
    from fourierdiffusion_amd import _rng
    from fourierdiffusion_amd.dataloaders.datamodules import TensorDatamodule
    from fourierdiffusion_amd.trainer import Trainer
    X = torch.arange(n * 3 * 2, dtype=torch.float32).reshape(n, 3, 2)          # X[i, 0, 0] = 6 i identifies the sample
    dm = TensorDatamodule(X_train=X, X_test=X[:4], batch_size=bs)
    model = _KeyDrawingModel(6)
    tr = Trainer(max_epochs=epochs, grad_exchange="torch", enable_progress_bar=False)
    torch.manual_seed(77)
    tr.fit(model, dm)
    per_epoch = len(model.seen) // epochs if rank == 0 else None
    torch.save({"seen": model.seen, "next_key": _rng.next_key(), "w": model.flat_parameters}, os.path.join(out_dir, f"rng_{rank}.pt"))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def test_world2_empty_slice_keeps_generators_and_shuffle_in_step(tmp_path):
    """n = 17, batch 8, world 2: the last global batch of every epoch has one sample, rank 1's slice is empty.  Over three
    epochs the two ranks must (a) cut the same permutation -- the union of their slices is every sample exactly once per
    epoch -- and (b) leave torch's global generator in the same state."""
    world, n, bs, epochs = 2, 17, 8, 3
    mp.spawn(_rng_fit_worker, args=(world, _free_port(), str(tmp_path), n, bs, epochs), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"rng_{r}.pt") for r in range(world)]
    assert res[0]["next_key"] == res[1]["next_key"], "torch's CPU generator drifted apart between the ranks"
    assert torch.equal(res[0]["w"], res[1]["w"])
    nb = (n + bs - 1) // bs
    # rank 0 ran nb steps per epoch, rank 1 nb - 1 (its slice of the last batch is empty)
    assert len(res[0]["seen"]) == epochs * nb and len(res[1]["seen"]) == epochs * (nb - 1)
    for e in range(epochs):
        ids = torch.cat(res[0]["seen"][e * nb:(e + 1) * nb] + res[1]["seen"][e * (nb - 1):(e + 1) * (nb - 1)])
        assert torch.equal(ids.sort().values, torch.arange(n, dtype=torch.float32) * 6), f"epoch {e}: samples duplicated or dropped"
    # epochs are shuffled differently
    assert not torch.equal(torch.cat(res[0]["seen"][:nb]), torch.cat(res[0]["seen"][nb:2 * nb]))
