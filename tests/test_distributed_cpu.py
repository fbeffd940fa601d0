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
