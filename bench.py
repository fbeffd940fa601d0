#!/usr/bin/env python3
"""bench.py -- headline benchmark of the fdiff hot path on MI355X.

metric  : sampled series/sec at (T=100, C=12)  [BASELINE.json]
workload: configs[1] -- "ecg (T=100, C=12) fourier_transform=true, default transformer
          (d_model=72, L=10, H=12, ff=2048), batch=512 on 1xMI355X", VP-SDE(0.1, 20), N=1000 reverse
          diffusion steps, synthetic prior / random-init weights (seed 42), inputs resident in HBM.
step    : ONE full reverse diffusion of one batch of 512 series (1000 x {score net + SDE step}).
N GPUs  : one process per GPU, each samples its own batch of 512 series (the sample batch is sharded;
          no data-path collective) -> weak scaling; value = N*512*K / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher environment re-executes itself under torch.distributed.run
(127.0.0.1 rendezvous on a free port), so both invocation styles print the same single JSON line from rank 0.

Other workloads (not the headline line; same JSON contract):
    --mode train                        BASELINE.json configs[2]: nasdaq-synth (T=252, C=6), default transformer, batch 64 per
                                        GPU, data-parallel: one bench step = one optimizer step (perturb + forward with dropout
                                        + backward in the bf16 MFMA kernels + flat RCCL gradient all-reduce + clip + fused AdamW)
    --workload mimic --scaling strong   BASELINE.json configs[3]: 4096 series (T=256, C=28), VE-SDE, 2000 predictor
                                        steps, the batch divided over the N ranks (strong scaling: total work fixed)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL across processes)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, L, H, F = 72, 10, 12, 2048        # default transformer (cmd/conf/score_model/default.yaml)
WORKLOADS = {
    # name: series shape, SDE, per-GPU batch of the weak-scaling line, reverse-diffusion steps, total batch of the strong line
    "ecg": dict(T=100, C=12, sde=("vp", 0.1, 20.0), batch=512, n_steps=1000, strong_total=4096,
                desc="BASELINE.json configs[1]: ecg-synth (T=100, C=12), default transformer (d_model=72, L=10, H=12, "
                     "ff=2048), VP-SDE, fourier_noise_scaling"),
    "mimic": dict(T=256, C=28, sde=("ve", 0.01, 2.0), batch=512, n_steps=2000, strong_total=4096,
                  desc="BASELINE.json configs[3]: mimiciii-synth (T=256, C=28), default transformer, VE-SDE(0.01, 2), "
                       "fourier_noise_scaling, 2000 predictor steps"),
    "nasdaq": dict(T=252, C=6, sde=("vp", 0.1, 20.0), batch=512, n_steps=1000, strong_total=4096,
                   desc="nasdaq-synth (T=252, C=6; the shape of BASELINE.json configs[2]), default transformer, VP-SDE, "
                        "fourier_noise_scaling"),
    "ecg187": dict(T=187, C=1, sde=("vp", 0.1, 20.0), batch=512, n_steps=1000, strong_total=4096,
                   desc="the reference's ECG dataset shape (T=187, C=1: src/fdiff/dataloaders/datamodules.py:194-201), default "
                        "transformer, VP-SDE, fourier_noise_scaling; persistent kernel specialised at run time (hiprtc)"),
    "long": dict(T=1024, C=16, sde=("vp", 0.1, 20.0), batch=64, n_steps=1000, strong_total=512,
                 desc="BASELINE.json configs[4]: synthetic long-horizon (T=1024, C=16), default transformer, VP-SDE, "
                      "fourier_noise_scaling, two launches per encoder layer (the series does not fit one workgroup)"),
}
T, CH = WORKLOADS["ecg"]["T"], WORKLOADS["ecg"]["C"]
TRAIN = dict(T=252, C=6, batch=64, desc="BASELINE.json configs[2]: nasdaq-synth (T=252, C=6), default transformer (d_model=72, "
             "L=10, H=12, ff=2048), VP-SDE, fourier_noise_scaling, dropout 0.1, AdamW + global-norm clip 1.0")
TRAIN_WORKLOADS = {
    "nasdaq": TRAIN,
    "ecg": dict(T=100, C=12, batch=64, desc="ecg-synth (T=100, C=12; the shape of BASELINE.json configs[1]), default transformer, "
                "VP-SDE, fourier_noise_scaling, dropout 0.1, AdamW + global-norm clip 1.0"),
}


def flops_per_series_forward(T=T, C=CH, D=D, L=L, F=F):
    """T*(L*(2*D*3D + 2*D*D + 4*D*F + 4*T*D) + 4*C*D) + 2*D*D  (padding flops do not count)."""
    return T * (L * (2 * D * 3 * D + 2 * D * D + 4 * D * F + 4 * T * D) + 4 * C * D) + 2 * D * D


def hbm_traffic_per_launch():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_hbm_traffic.json,
    collected with separate rocprofv3 --pmc runs of this same command; FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md).  (None, None) when no measurement is committed; the second element names the source."""
    for name in ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                v = json.load(f).get("hbm_bytes_per_launch")
            if v is not None:
                return v, f"profiles/{name}: separate rocprofv3 --pmc passes of this command, NOT measured in this run"
        except Exception:
            pass
    return None, None


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: one rank per GPU under torch.distributed.run, same arguments."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def cpu_baseline(batch: int, T: int, CH: int, n_timed: int = 3):
    """The reference's CPU path = the same torch-op sequence it executes (nn.TransformerEncoder eval fast path +
    diag_embed/matmul scheduler step), timed on this host's cores by oracle/torch_cpu_baseline.py on a bounded
    sample (a few reverse-diffusion steps, extrapolated x1000: every step costs the same)."""
    from oracle import torch_cpu_baseline as cb
    return cb.time_sampler_steps(batch=batch, T=T, C=CH, d_model=D, num_layers=L, n_head=H, n_timed=n_timed)


class BoardSampler:
    """Board power and shader clock of the first amdgpu device as its hwmon reports them (sysfs: power1_average / power1_input in
    microwatts, freq1_input in Hz), sampled twice a second while the timed region runs.  The persistent kernel is power-limited
    (DESIGN.md section 3.3): the record carries what the board drew and clocked beside the time.  None when sysfs has no such node."""

    def __init__(self):
        import glob
        self.nodes = None
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            pw = [p for p in (os.path.join(hw, "power1_average"), os.path.join(hw, "power1_input")) if os.path.exists(p)]
            fq = os.path.join(hw, "freq1_input")
            if pw:
                self.nodes = (pw[0], fq if os.path.exists(fq) else None)
                break
        self.samples = []
        self._stop = None
        self._thread = None

    def _read(self):
        try:
            w = int(open(self.nodes[0]).read().strip()) / 1e6
            f = int(open(self.nodes[1]).read().strip()) / 1e6 if self.nodes[1] else None
            self.samples.append((w, f))
        except (OSError, ValueError):
            pass

    def start(self):
        if not self.nodes:
            return
        import threading
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                self._read()
                self._stop.wait(0.5)
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def stop(self):
        if not self._thread:
            return None
        self._stop.set()
        self._thread.join(timeout=2)
        if not self.samples:
            return None
        ws = [w for w, _ in self.samples]
        fs = [f for _, f in self.samples if f]
        out = {"samples": len(ws), "power_w_mean": round(sum(ws) / len(ws), 1), "power_w_max": round(max(ws), 1), "source": self.nodes[0]}
        if fs:
            out["sclk_mhz_mean"] = round(sum(fs) / len(fs), 1)
            out["sclk_mhz_min"] = round(min(fs), 1)
        return out


def secondary_rows():
    """The rows of the path that are not the headline (the training step, the other BASELINE.json shapes, the
    HBM-bound transforms), each measured by its own process AFTER the headline's timed region and reported inside the same
    JSON line, so that the driver's single default run sees them too.  None of this touches `value` / `ms_per_step`.
    A row that fails reports {"error": ...}; the headline is printed regardless."""
    import subprocess
    me = os.path.abspath(__file__)
    quick = bool(os.environ.get("FDIFF_BENCH_SECONDARY_QUICK"))          # (tests: a few steps per row)
    nd = ["--diffusion-steps", "20"] if quick else []           # (otherwise each workload's own step count: 1000 / 2000)
    common = ["--no-cpu-baseline", "--no-secondary", "--gpus", "1"]
    rows = {
        "train_nasdaq_T252_B64": [me, "--mode", "train", "--train-workload", "nasdaq", "--steps", "5" if quick else "100"],
        "train_ecg_T100_B64": [me, "--mode", "train", "--train-workload", "ecg", "--steps", "5" if quick else "100"],
        "sample_long_T1024_B64": [me, "--workload", "long", "--steps", "2", "--warmup", "1"] + nd,
        "sample_mimic_T256_B512": [me, "--workload", "mimic", "--steps", "2", "--warmup", "1"] + nd,
        "sample_nasdaq_T252_B512": [me, "--workload", "nasdaq", "--steps", "2", "--warmup", "1"] + nd,
        "sample_ecg187_T187_B512": [me, "--workload", "ecg187", "--steps", "2", "--warmup", "1"] + nd,
    }
    keep = ("metric", "value", "unit", "steps", "ms_per_step", "score_net_step_ms", "dtype", "achieved_tflops_whole_step")
    out = {}
    for name, cmd in rows.items():
        try:
            r = subprocess.run([sys.executable] + cmd + common, capture_output=True, text=True, timeout=300)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            j = json.loads(line)
            row = {k: j[k] for k in keep if k in j}
            row["workload"] = j["config"]["workload"]
            if j.get("roofline"):
                row["roofline"] = {k: j["roofline"][k] for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_kernel_us",
                                                                 "shader_clock_mhz", "frac_at_nominal_clock") if k in j["roofline"]}
            out[name] = row
        except Exception as e:
            out[name] = {"error": repr(e)[:300]}
    try:   # the reference's default sampling run through the Python boundary (10 000 samples in batches of 200, 1000 steps; PCIe-inclusive)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "api_default_run.py"), "100", "12", "2000" if quick else "10000",
                            "100" if quick else "1000"], capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("DiffusionSampler")][-1]
        out["api_default_run_T100_C12"] = {"what": line, "series_per_s": float(line.split(" = ")[1].split(" series/s")[0])}
    except Exception as e:
        out["api_default_run_T100_C12"] = {"error": repr(e)[:300]}
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "hbm_kernels_bench.py"), "--json"] +
                           (["--quick"] if quick else []), capture_output=True, text=True, timeout=300)
        out["hbm_kernels"] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as e:
        out["hbm_kernels"] = {"error": repr(e)[:300]}
    return out


def rank_report(dist, rank, world, dev_index):
    """FDIFF_BENCH_REPORT_RANKS=1 (scripts/scale_check.sh): every rank's device binding and Philox counter base, gathered on
    rank 0 -- one process per GPU and disjoint noise streams are checked on the real node, not assumed."""
    if not os.environ.get("FDIFF_BENCH_REPORT_RANKS"):
        return None
    from fourierdiffusion_amd import _rng
    props = torch.cuda.get_device_properties(dev_index)
    me = {"rank": rank, "device": dev_index, "uuid": str(getattr(props, "uuid", "")), "philox_base": _rng.base_offset(),
          "pid": os.getpid()}
    if dist is None:
        return [me]
    out = [None] * world
    dist.all_gather_object(out, me)
    return out


def main_train(args, rank, local_rank, world):
    """configs[2]: batch-sharded training.  Every rank holds 64 synthetic series (weak scaling); a step = one optimizer step."""
    TR = TRAIN_WORKLOADS[args.train_workload]
    T, CH = TR["T"], TR["C"]
    B = args.batch or TR["batch"]
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("FDIFF_BENCH_BACKEND", "nccl")
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist_mod.init_process_group(backend="nccl", device_id=dev)
        else:
            dist_mod.init_process_group(backend=backend)
        dist = dist_mod
    from fourierdiffusion_amd import _C, _rng
    from fourierdiffusion_amd.models.score_models import ScoreModule
    from fourierdiffusion_amd.optim import FusedAdamW
    from fourierdiffusion_amd.parallel import DistEnv, GradExchange
    from fourierdiffusion_amd.schedulers.sde import VPScheduler
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    torch.manual_seed(42)
    _rng.set_rank(rank)
    sch = VPScheduler(beta_min=0.1, beta_max=20.0, fourier_noise_scaling=True)
    sch.set_noise_scaling(T)
    model = ScoreModule(n_channels=CH, max_len=T, noise_scheduler=sch, fourier_noise_scaling=True, d_model=D, num_layers=L,
                        n_head=H).to(dev)
    model.train_precision = "bf16" if args.precision == "bf16" else "fp32"
    model.train()
    opt = FusedAdamW(model, lr=1e-3, max_grad_norm=1.0)
    ex = GradExchange(DistEnv(rank, local_rank, world), backend="rccl" if backend == "nccl" else "torch")
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    X = torch.randn(B, T, CH, generator=g).to(dev)          # each rank its own shard of the global batch
    ctx, _ = model._engine()
    lib = _C.lib()

    def one_step(i):
        model.zero_grad()
        loss = model.training_step(DiffusableBatch(X=X), i)
        ex.all_reduce_mean(model.grads)
        opt.step()
        return loss

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    steps = args.steps if args.steps != 2 else 50           # (2 is the sampling default: an optimizer step is 1000x shorter)
    warmup = max(args.warmup, 3)
    for i in range(warmup):
        one_step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = one_step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(loss).all() and torch.isfinite(model.flat_parameters).all(), "training produced non-finite values"
    # Roofline leg: the five per-layer kernels are launched 50 times per step between other kernels of the same stream, and an event
    # record is a barrier packet of a few us.  Measured on MI355X (profiles/r04_train_prof_brackets.txt): with EVERY launch bracketed
    # the per-kernel averages agree with rocprofv3 but the step takes 14 % longer; with a sampled subset (every 7th / 23rd launch) the
    # step is nearly undisturbed but a bracket whose neighbours are not bracketed mis-times its kernel by up to 40 us (k_tr_ffn_bwd
    # 82 instead of 42 us, k_tr_attn_bwd 69 instead of 97).  So the timed region above runs without brackets, and PROF_STEPS more
    # optimizer steps of the same loop follow with every launch bracketed: fd_prof_end names the kernel with the largest total time.
    PROF_STEPS = 5
    _C.check(lib.fd_prof_begin(ctx), ctx)
    _C.check(lib.fd_prof_stride(ctx, int(os.environ.get("FDIFF_BENCH_PROF_STRIDE", "1"))), ctx)
    for i in range(PROF_STEPS):
        one_step(steps + i)
    barrier()
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ranks_info = rank_report(dist, rank, world, dev_index)
    if rank == 0:
        fwd_flops = flops_per_series_forward(T=T, C=CH)
        out = {
            "metric": f"training series/sec (T={T}, C={CH})", "value": world * B * steps / elapsed, "unit": "series/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if model.train_mode_effective == "bf16" else "f32",
            "data": "synthetic (N(0,1) series per rank, random-init weights seed 42)",
            "config": {"workload": TR["desc"] + f", batch={B}/GPU, one optimizer step per bench step",
                       "global_batch": world * B, "seq_len": T, "parallelism": f"data parallel x{world}, flat RCCL all-reduce"},
            "achieved_tflops_whole_step": 3 * fwd_flops * world * B * steps / elapsed / 1e12,
        }
        avg_us, cnt, flops = C.c_double(0), C.c_int(0), C.c_double(0)
        name = C.create_string_buffer(128)
        roof = None
        if lib.fd_prof_end(ctx, name, C.byref(avg_us), C.byref(cnt), C.byref(flops)) == 0 and cnt.value > 0:
            peak = 2500.0 if model.train_mode_effective == "bf16" else 157.3
            ach = flops.value / (avg_us.value * 1e-6) / 1e12
            roof = {"bound": "mfma", "kernel": name.value.decode(), "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                    "frac": ach / peak, "traffic": None, "avg_kernel_us": avg_us.value, "launches_timed": cnt.value,
                    "flops_per_launch": flops.value,
                    "measured": f"{PROF_STEPS} further optimizer steps of the same loop right after the timed region, every launch of the "
                                "five per-layer kernels bracketed by HIP events on its stream (brackets inside the timed region "
                                "lengthen the step by up to 14 %; sampled brackets mis-time their kernels)"}
        out["roofline"] = roof
        if ranks_info is not None:
            out["ranks"] = ranks_info
        if world == 1 and not args.no_cpu_baseline:
            try:
                from oracle import torch_cpu_baseline as cb
                out["cpu_baseline"] = cb.time_train_steps(batch=B, T=T, C=CH, d_model=D, num_layers=L, n_head=H)
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", default="sample", choices=["sample", "train"])
    ap.add_argument("--workload", default="ecg", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--batch", type=int, default=None, help="series per GPU (weak) / in total (strong)")
    ap.add_argument("--diffusion-steps", type=int, default=None)
    ap.add_argument("--precision", default=os.environ.get("FDIFF_PRECISION", "bf16"), choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-workload", default="nasdaq", choices=sorted(TRAIN_WORKLOADS))
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the other rows of the path that the default one-GPU run measures after its timed region")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.mode == "train":
        return main_train(args, rank, local_rank, world)
    wl = WORKLOADS[args.workload]
    T, CH = wl["T"], wl["C"]
    # one rank per GPU; the modulo only matters for the single-GPU rehearsal of the multi-rank path
    # (FDIFF_BENCH_BACKEND=gloo, tests/test_gpu_entrypoints.py), where two ranks share cuda:0
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    backend = os.environ.get("FDIFF_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist_mod.init_process_group(backend="nccl", device_id=dev)
        else:
            dist_mod.init_process_group(backend=backend)
        dist = dist_mod

    from fourierdiffusion_amd import _C, _rng
    from fourierdiffusion_amd.models.score_models import ScoreModule
    from fourierdiffusion_amd.parallel import shard_range
    from fourierdiffusion_amd.schedulers.sde import VEScheduler, VPScheduler

    torch.manual_seed(42)
    _rng.set_rank(rank)
    kind, p0, p1 = wl["sde"]
    if kind == "vp":
        sch = VPScheduler(beta_min=p0, beta_max=p1, fourier_noise_scaling=True)
    else:
        sch = VEScheduler(sigma_min=p0, sigma_max=p1, fourier_noise_scaling=True)
    sch.set_noise_scaling(T)
    model = ScoreModule(n_channels=CH, max_len=T, noise_scheduler=sch, fourier_noise_scaling=True, d_model=D,
                        num_layers=L, n_head=H).to(dev)
    model.precision = args.precision
    model.eval()
    N = args.diffusion_steps or wl["n_steps"]
    if args.scaling == "weak":
        B = args.batch or wl["batch"]                     # per GPU, fixed as the number of GPUs grows
        total_series = world * B
    else:
        total_series = args.batch or wl["strong_total"]   # fixed total, divided over the ranks
        lo, hi = shard_range(total_series, rank, world)
        B = hi - lo
    sch.set_timesteps(N)
    ctx, h = model._engine()
    lib = _C.lib()
    ts_arr = (C.c_float * N)(*sch.timesteps.tolist())
    prm = sch._c_params()
    G = sch.G_on(dev)
    mode = _C.FD_MODE_BF16 if args.precision == "bf16" else _C.FD_MODE_F32
    stream = torch.cuda.current_stream(dev).cuda_stream
    X = torch.empty(B, T, CH, device=dev)

    def one_step(i):
        # prior (on device) + N x {score net, SDE step}: all enqueued on `stream`, no host sync inside
        key, off = _rng.stream()
        _C.check(lib.fd_prior_sample(ctx, C.byref(prm), G.data_ptr(), None, key, off, X.data_ptr(), B, T, CH,
                                     stream), ctx)
        key, off = _rng.stream()
        _C.check(lib.fd_sampler_run(h, C.byref(prm), G.data_ptr(), ts_arr, N, float(sch.step_size), X.data_ptr(),
                                    None, key, off, B, mode, stream), ctx)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        one_step(i)
    barrier()
    prof = True
    _C.check(lib.fd_prof_begin(ctx), ctx)
    sampler = BoardSampler() if rank == 0 else None      # (a thread that reads two sysfs nodes twice a second: no device work)
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    board = sampler.stop() if sampler else None
    assert torch.isfinite(X).all(), "sampler produced non-finite values"
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ranks_info = rank_report(dist, rank, world, dev_index)
    if rank == 0:
        series = total_series * args.steps
        value = series / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        fwd_flops = flops_per_series_forward(T=T, C=CH)      # per series per score-net forward
        out = {
            "metric": f"sampled series/sec (T={T}, C={CH})",
            "value": value,
            "unit": "series/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "score_net_step_ms": ms_per_step / N,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.precision if args.precision == "bf16" else "f32",
            "data": "synthetic (Philox prior, random-init weights seed 42)",
            "config": {"workload": wl["desc"] + (f", batch={B}/GPU" if args.scaling == "weak" else
                                                 f", {total_series} series divided over the ranks")
                                   + f", {N} reverse-diffusion steps per bench step",
                       "global_batch": total_series, "seq_len": T, "parallelism": f"sample-batch shard x{world}"},
            "achieved_tflops_whole_step": fwd_flops * N * total_series * args.steps / elapsed / 1e12,
        }
        roof = None
        if prof:
            avg_us = C.c_double(0)
            cnt = C.c_int(0)
            flops = C.c_double(0)
            name = C.create_string_buffer(128)
            if lib.fd_prof_end(ctx, name, C.byref(avg_us), C.byref(cnt), C.byref(flops)) == 0 and cnt.value > 0:
                peak = 2500.0 if args.precision == "bf16" else 157.3     # dense MFMA peak, MI355X_MICROARCH.md
                ach = flops.value / (avg_us.value * 1e-6) / 1e12
                traffic, traffic_src = hbm_traffic_per_launch() if args.workload == "ecg" and B == 512 and N == 1000 \
                    else (None, None)
                roof = {"bound": "mfma", "kernel": name.value.decode(), "instantiation": model.plan(B)[0],
                        "achieved": ach, "peak": peak,
                        "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src,
                        "avg_kernel_us": avg_us.value, "launches_timed": cnt.value,
                        "flops_per_launch": flops.value}
                # the shader clock the persistent kernel ran at (its own shader-clock counter against the wall clock): the chip is
                # power-limited on this kernel and boxes of the pool differ by +-2.5 % -- the record carries the clock beside the time
                mhz = C.c_double(0)
                if lib.fd_prof_shader_clock_mhz(ctx, C.byref(mhz)) == 0 and mhz.value > 0:
                    roof["shader_clock_mhz"] = round(mhz.value, 1)
                    # the same cycles at the nominal 2.4 GHz: comparable across boxes and rounds (the delivered `frac` above is not --
                    # the same library gave 0.308-0.322 on boxes that clocked 2266-2354 MHz under the power cap)
                    roof["frac_at_nominal_clock"] = roof["frac"] * 2400.0 / mhz.value
                if board:
                    roof["board"] = board
        out["roofline"] = roof
        if ranks_info is not None:
            out["ranks"] = ranks_info
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(B, T, CH)
            except Exception as e:   # the GPU result must still be reported
                out["cpu_baseline"] = {"error": repr(e)}
        if world == 1 and args.workload == "ecg" and not args.no_secondary:
            del X
            torch.cuda.empty_cache()
            out["secondary"] = secondary_rows()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
