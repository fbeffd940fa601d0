"""Time the hot path at the other BASELINE.json shapes (not bench lines: DESIGN.md section 4 quotes them).
usage: python scripts/shape_bench.py [sample|train] NAME B [diffusion_steps]
`train` also runs data-parallel under torch.distributed.run (BASELINE.json configs[2]: nasdaq, batch 64 per GPU x 8):
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/shape_bench.py train nasdaq 64
one process per GPU, the gradient exchange is ONE RCCL all-reduce of the flat fp32 gradient buffer per optimizer step."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = {"ecg": (100, 12), "nasdaq": (252, 6), "mimic": (256, 28), "long": (1024, 16), "ecg187": (187, 1), "mimic24": (24, 40),
          "nasa": (134, 10),
          # the shapes the reference's datamodules produce (datamodules.py:194-201, 404-410, 471-476; mimiciii.yaml:7)
          "nasdaq5": (252, 5), "nasa251": (251, 4), "nasa134": (134, 5),
          # droughts: one year of 18 daily indicators minus the five the datamodule removes (datamodules.py:528-537)
          "droughts": (365, 13)}


def flops_fwd(T, Cn, D=72, L=10, F=2048):
    return T * (L * (2 * D * 3 * D + 2 * D * D + 4 * D * F + 4 * T * D) + 4 * Cn * D) + 2 * D * D


def main():
    what, name, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    T, Cn = SHAPES[name]
    from fourierdiffusion_amd import _C, _rng
    from fourierdiffusion_amd.models.score_models import ScoreModule
    from fourierdiffusion_amd.schedulers.sde import VPScheduler
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    torch.manual_seed(0)
    from fourierdiffusion_amd.parallel import GradExchange, init_process_group
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    denv = init_process_group(os.environ.get("FDIFF_BENCH_BACKEND"))
    _rng.set_rank(denv.rank)
    sch = VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True)
    sch.set_noise_scaling(T)
    m = ScoreModule(n_channels=Cn, max_len=T, noise_scheduler=sch, fourier_noise_scaling=True, d_model=72, num_layers=10,
                    n_head=12).to(dev)
    m.precision = os.environ.get("FDIFF_PRECISION", "bf16")
    m.train_precision = os.environ.get("FDIFF_TRAIN_PRECISION", "bf16")
    if os.environ.get("FDIFF_DROPOUT"):          # (experiments: 0 = what the dropout decisions cost)
        m.dropout = float(os.environ["FDIFF_DROPOUT"])
    if what == "sample":
        m.eval()
        sch.set_timesteps(N)
        ctx, h = m._engine()
        lib = _C.lib()
        ts = (C.c_float * N)(*sch.timesteps.tolist())
        prm = sch._c_params()
        G = sch.G_on(dev)
        X = torch.randn(B, T, Cn, device=dev)
        mode = _C.FD_MODE_BF16 if m.precision == "bf16" else _C.FD_MODE_F32
        st = torch.cuda.current_stream(dev).cuda_stream

        def run():
            key, off = _rng.stream()
            _C.check(lib.fd_sampler_run(h, C.byref(prm), G.data_ptr(), ts, N, float(sch.step_size), X.data_ptr(), None, key, off,
                                        B, mode, st), ctx)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N
        tf = B * flops_fwd(T, Cn) / dt / 1e12
        print(f"{name} sample B={B} T={T} C={Cn} {m.precision}: {1e3 * dt:.3f} ms per diffusion step, "
              f"{tf:.1f} TFLOP/s algorithmic = {tf / 2500:.4f} of the bf16 peak, {B / (dt * 1000):.1f} series/s at N=1000 "
              f"[N={N}, FDIFF_MEGA_JIT={os.environ.get('FDIFF_MEGA_JIT', 'auto')}; {m.plan(B)[0].split(' S=')[0]}]")
    else:
        from fourierdiffusion_amd.optim import FusedAdamW
        m.train()
        opt = FusedAdamW(m, lr=1e-3)
        X = torch.randn(B, T, Cn, device=dev)
        ex = GradExchange(denv, backend="rccl" if os.environ.get("FDIFF_BENCH_BACKEND", "nccl") == "nccl" else "torch")

        def step():
            m.zero_grad()
            loss = m.training_step(DiffusableBatch(X=X), 0)
            ex.all_reduce_mean(m.grads)
            opt.step()
            return loss
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        nrep = int(os.environ.get("FDIFF_BENCH_NREP", "100"))
        t0 = time.perf_counter()
        for _ in range(nrep):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nrep
        if denv.world > 1:
            import torch.distributed as dist
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        if denv.rank == 0:
            print(f"{name} train ({m.train_mode_effective}) B={B}/GPU x {denv.world} T={T} C={Cn}: {1e3 * dt:.3f} ms per optimizer step "
                  f"(fwd+bwd+all-reduce+AdamW), {3 * denv.world * B * flops_fwd(T, Cn) / dt / 1e12:.2f} TFLOP/s algorithmic, "
                  f"{denv.world * B / dt:.0f} series/s")


main()
