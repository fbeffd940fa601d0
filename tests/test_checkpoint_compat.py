"""Lightning-checkpoint compatibility with the reference, CPU part: a checkpoint in the reference's layout
-- written by the reference itself (oracle/make_golden.py gen_ckpt -> tests/golden/reference_tiny.ckpt; the scheduler object
is pickled inside under fdiff.schedulers.sde.VPScheduler) -- loads here, and a checkpoint written here names the reference's
class paths and loads in the reference (checked in the build container, where /root/reference exists)."""
import os
import pickletools
import subprocess
import sys
import zipfile

import numpy as np
import pytest
import torch

from oracle import weights as W
from oracle.make_golden import CFG_TINY

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CKPT = os.path.join(ROOT, "tests", "golden", "reference_tiny.ckpt")


def test_reference_checkpoint_loads_through_the_alias():
    import fdiff  # noqa: F401  (the alias package: fdiff.schedulers.sde -> fourierdiffusion_amd.schedulers.sde)
    from fourierdiffusion_amd.models.score_models import ScoreModule
    from fourierdiffusion_amd.schedulers.sde import VPScheduler
    ck = torch.load(CKPT, map_location="cpu", weights_only=False)
    assert ck["pytorch-lightning_version"] and ck["hparams_name"] == "kwargs"
    sch = ck["hyper_parameters"]["noise_scheduler"]
    assert type(sch) is VPScheduler and sch.beta_0 == 0.1 and sch.beta_1 == 20.0 and sch.noise_scaling is True
    m = ScoreModule.load_from_checkpoint(CKPT)
    assert (m.n_channels, m.max_len, m.d_model, m.num_layers, m.n_head) == (3, 20, 8, 2, 4)
    assert isinstance(m.noise_scheduler, VPScheduler) and m.noise_scheduler.G.shape == (20,)
    want = W.make_state_dict(CFG_TINY["C"], CFG_TINY["T"], CFG_TINY["D"], CFG_TINY["L"], seed=1234)
    got = m.state_dict()
    assert sorted(got) == sorted(want)
    for k, v in want.items():
        if k == "pos_encoder.pe.weight":
            continue        # (rows above the max_norm are renormalised on load, as the reference does on first use)
        np.testing.assert_array_equal(got[k].cpu().numpy(), v, err_msg=k)


def _pickled_globals(path):
    with zipfile.ZipFile(path) as z:
        name = [n for n in z.namelist() if n.endswith("data.pkl")][0]
        ops = pickletools.genops(z.read(name))
        out, strings = set(), []
        for op, arg, _ in ops:
            if op.name in ("SHORT_BINUNICODE", "BINUNICODE", "UNICODE"):
                strings.append(arg)
            elif op.name == "STACK_GLOBAL" and len(strings) >= 2:
                out.add(f"{strings[-2]}.{strings[-1]}")
            elif op.name == "GLOBAL":
                out.add(arg.replace(" ", "."))
        return out


def test_checkpoint_written_here_names_reference_paths_and_round_trips(tmp_path):
    import fdiff  # noqa: F401
    from fourierdiffusion_amd.models.score_models import ScoreModule
    m = ScoreModule.load_from_checkpoint(CKPT)
    path = tmp_path / "epoch=0-val_loss=0.50.ckpt"
    m.save_checkpoint(path, epoch=0, global_step=7)
    names = _pickled_globals(path)
    assert "fdiff.schedulers.sde.VPScheduler" in names
    assert not [n for n in names if n.startswith("fourierdiffusion_amd")], names      # nothing the reference cannot import
    again = ScoreModule.load_from_checkpoint(path)
    for k, v in m.state_dict().items():
        np.testing.assert_array_equal(again.state_dict()[k].cpu().numpy(), v.cpu().numpy(), err_msg=k)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    for key in ("epoch", "global_step", "pytorch-lightning_version", "state_dict", "hparams_name", "hyper_parameters"):
        assert key in ck


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference only exists in the build container")
def test_reference_loads_a_checkpoint_written_here(tmp_path):
    import fdiff  # noqa: F401
    from fourierdiffusion_amd.models.score_models import ScoreModule
    path = tmp_path / "engine.ckpt"
    ScoreModule.load_from_checkpoint(CKPT).save_checkpoint(path)
    script = f"""
import sys, inspect
sys.argv = ["x"]
sys.path.insert(0, {ROOT!r})
import numpy as np, torch
import oracle.make_golden as G
from oracle import weights as W
R = G.import_reference()
ck = torch.load({str(path)!r}, map_location="cpu", weights_only=False)
sch = ck["hyper_parameters"]["noise_scheduler"]
assert type(sch).__module__ == "fdiff.schedulers.sde" and inspect.getfile(type(sch)).startswith("/root/reference"), type(sch)
hp = {{k: v for k, v in ck["hyper_parameters"].items() if k in inspect.signature(R.sm.ScoreModule.__init__).parameters}}
m = R.sm.ScoreModule(**hp)
m.load_state_dict(ck["state_dict"], strict=True)
m.eval()
X = W.randn("score_x_tiny", (3, 20, 3), 2); t = W.uniform("score_t_tiny", (3,), 2, 1e-5, 1.0)
with torch.no_grad():
    out = m(R.dc.DiffusableBatch(X=torch.from_numpy(X), y=None, timesteps=torch.from_numpy(t))).numpy()
g = np.load({os.path.join(ROOT, "tests", "golden", "score_forward.npz")!r})
np.testing.assert_allclose(out, g["fast_tiny"], atol=5e-6, rtol=0)
print("reference-side load ok")
"""
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "reference-side load ok" in r.stdout, r.stderr[-2000:]
