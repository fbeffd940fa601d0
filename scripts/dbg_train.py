import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from oracle import weights as W
from oracle.make_golden import CFG_TINY, CFG_DEFAULT
from tests.gpu_util import make_model, dev
from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
from fourierdiffusion_amd.utils.losses import get_sde_loss_fn
for name, cfg, B in (("default", CFG_DEFAULT, 6), ("tiny", CFG_TINY, 5)):
    X = W.randn("x", (B, cfg["T"], cfg["C"]), 3); z = W.randn("z", (B, cfg["T"], cfg["C"]), 3); t = W.uniform("t", (B,), 3, 0.05, 1.0)
    m, sch, _ = make_model(cfg, precision="bf16")
    m.dropout = 0.0
    m.zero_grad()
    print(name, "forward...", flush=True)
    fn = get_sde_loss_fn(sch, train=True)
    l = fn(m, DiffusableBatch(X=dev(X), timesteps=dev(t)), noise=dev(z), backward=False)
    torch.cuda.synchronize(); print("  loss", l.item(), flush=True)
    l = fn(m, DiffusableBatch(X=dev(X), timesteps=dev(t)), noise=dev(z), backward=True)
    torch.cuda.synchronize(); print("  bwd ok", float(m.grads.abs().max()), flush=True)
