"""GPU: the scripted multi-GPU day (scripts/scale_check.sh) cannot rot -- it runs here at N = 1 with a few steps per mode
(sampling weak, mimic strong, training), each line checked for the bench contract's keys, one device per rank and disjoint
Philox ranges; and the engine's RCCL binding is the process's ONE librccl image (torch's bundled copy under the Python host)."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_scale_check_script_runs_at_one_gpu():
    env = dict(os.environ, SCALE_CHECK_FAST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "scale_check.sh"), "1"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    oks = [l for l in r.stdout.splitlines() if l.strip().startswith("ok")]
    assert len(oks) == 3, r.stdout[-3000:]


def _rccl_images():
    out = set()
    with open("/proc/self/maps") as f:
        for line in f:
            path = line.split(None, 5)[-1].strip() if line.count(" ") >= 5 else ""
            if "librccl" in os.path.basename(path):
                out.add(os.path.realpath(path))
    return out


def _engine_rccl_path():
    from fourierdiffusion_amd import _C
    buf = C.create_string_buffer(4096)
    rc = _C.lib().fd_comm_rccl_path(buf, 4096)
    assert rc == 0, f"fd_comm_rccl_path failed ({rc}): librccl not loadable"
    return os.path.realpath(buf.value.decode())


def test_engine_and_torch_share_one_rccl_image():
    """fd_comm.hip dlopens librccl beside torch's bundled torch/lib/librccl.so.  The binding takes an
    already mapped image (RTLD_NOLOAD, matched by SONAME), so a process holds one copy -- asserted through /proc/self/maps."""
    import torch  # noqa: F401  (loads torch/lib/librccl.so through libtorch_hip.so)
    before = _rccl_images()
    mine = _engine_rccl_path()
    after = _rccl_images()
    assert len(after) == 1, f"two RCCL images in one process: {after}"
    assert mine in after
    if before:
        assert before == after, (before, after)
        tl = os.path.realpath(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        assert mine == tl, (mine, tl)


@pytest.mark.gpu
def test_one_rccl_image_after_comm_init():
    """Same assertion after a real (1-rank) communicator was created and an all-reduce ran on the device."""
    import torch
    from fourierdiffusion_amd import _C
    dev = torch.device("cuda:0")
    ctx = _C.ctx(dev)
    buf = (C.c_ubyte * _C.FD_COMM_ID_BYTES)()
    assert _C.lib().fd_comm_unique_id(buf) == 0
    _C.check(_C.lib().fd_comm_init(ctx, 0, 1, buf), ctx)
    g = torch.arange(1000, dtype=torch.float32, device=dev)
    _C.check(_C.lib().fd_allreduce_grads(ctx, g.data_ptr(), g.numel(), 0.5, _C.stream_of(g)), ctx)
    torch.cuda.synchronize()
    assert torch.equal(g.cpu(), torch.arange(1000, dtype=torch.float32) * 0.5)
    _C.check(_C.lib().fd_comm_destroy(ctx), ctx)
    assert len(_rccl_images()) == 1 and _engine_rccl_path() in _rccl_images()
