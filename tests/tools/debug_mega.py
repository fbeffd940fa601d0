"""Debug aid: compare the persistent kernel against the oracle layer by layer (FDIFF_MEGA_LAYERS=k)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import fdiff_oracle as O, weights as W
from oracle.make_golden import CFG_DEFAULT, CFG_TINY, CFG_ODD
from tests.gpu_util import make_model, dev, host
from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch

name = sys.argv[1] if len(sys.argv) > 1 else "default"
cfgs = {"default": CFG_DEFAULT, "tiny": CFG_TINY, "odd": CFG_ODD,
        "nasdaq": dict(T=252, C=6, D=72, L=10, H=12), "mimic": dict(T=256, C=28, D=72, L=10, H=12),
        "long": dict(T=1024, C=16, D=72, L=10, H=12),
        "drought": dict(T=365, C=1, D=72, L=3, H=12), "cls": dict(T=128, C=5, D=60, L=3, H=12),
        "cls_long": dict(T=400, C=7, D=60, L=3, H=12), "wide_head": dict(T=300, C=4, D=24, L=2, H=2),
        "t2048": dict(T=2048, C=4, D=72, L=1, H=12), "ecg187": dict(T=187, C=1, D=72, L=10, H=12),
        "mimic24": dict(T=24, C=40, D=72, L=10, H=12), "nasa": dict(T=134, C=10, D=72, L=10, H=12)}
cfg = cfgs[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nl = int(os.environ.get("FDIFF_MEGA_LAYERS", cfg["L"]))
m, _, sd = make_model(cfg, precision="bf16")
X = W.randn("dbg_x", (B, cfg["T"], cfg["C"]), 2)
t = W.uniform("dbg_t", (B,), 2, 1e-5, 1.0)
m.eval()
out = host(m(DiffusableBatch(X=dev(X), timesteps=dev(t))))
sd2 = {k: v for k, v in sd.items() if not k.startswith("backbone.layers.") or int(k.split(".")[2]) < nl}
ref = O.score_forward(sd2, X, t, cfg["H"])
err = np.abs(out - ref)
print(f"{name} layers={nl} B={B}: nan={np.isnan(out).sum()} max_err={np.nanmax(err):.4e} ref_max={np.abs(ref).max():.3f} "
      f"rms_rel={np.sqrt(np.nanmean(err**2))/np.sqrt((ref**2).mean()):.4e}")
if np.isnan(out).any():
    idx = np.argwhere(np.isnan(out))
    print("first nan idx", idx[:5], "count per batch", [int(np.isnan(out[b]).sum()) for b in range(B)])
else:
    w = np.unravel_index(np.argmax(err), err.shape)
    print("worst at", w, out[w], ref[w])
