"""CPU: the C-ABI shared library loads and exports every symbol include/fdiff_hip.h declares
(no compute calls without a GPU), and the engine-defined parameter layout is sane."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fdiff_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fd_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ("fd_rfft_pack", "fd_irfft_unpack", "fd_sde_step", "fd_score_forward", "fd_score_backward",
                 "fd_sampler_run", "fd_adamw_step", "fd_allreduce_grads"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from fourierdiffusion_amd import _C
    assert os.path.exists(_C.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_C.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    # and the Python binding table covers the header exactly
    assert sorted(_C.EXPORTED_SYMBOLS) == declared_symbols()
    assert _C.lib().fd_version() >= 100


def test_layout_matches_reference_state_dict_keys():
    from fourierdiffusion_amd import _C
    dims = _C.model_dims(12, 100, 72, 12, 10)
    layout, total = _C.score_layout(dims)
    names = [e[0] for e in layout]
    assert names[:8] == ["pos_encoder.embedding.weight", "time_encoder.W", "time_encoder.dense.weight",
                         "time_encoder.dense.bias", "embedder.weight", "embedder.bias", "unembedder.weight",
                         "unembedder.bias"]
    assert names[8] == "backbone.layers.0.self_attn.in_proj_weight"
    assert len(names) == 8 + 12 * 10
    assert sum(e[2] for e in layout) == 3197744
    assert total >= 3197744
    for _, off, _, _, _ in layout:
        assert off % 4 == 0                                # 16-byte aligned tensors
    frozen = [e[0] for e in layout if not e[4]]
    assert frozen == ["time_encoder.W"]                    # requires_grad=False (transformer.py:72-74)
    # odd sizes get padded, not overlapped
    layout2, total2 = _C.score_layout(_C.model_dims(3, 21, 10, 2, 1))
    ends = [off + n for _, off, n, _, _ in layout2]
    for (_, off, _, _, _), prev_end in zip(layout2[1:], ends[:-1]):
        assert off >= prev_end
    assert total2 >= ends[-1]


def test_bad_dims_rejected():
    from fourierdiffusion_amd import _C
    with pytest.raises(_C.FdError):
        _C.score_layout(_C.model_dims(3, 20, 10, 3, 1))    # d_model % n_head != 0


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU instead of computing on the CPU."""
    import torch
    from fourierdiffusion_amd import _C
    from fourierdiffusion_amd.models.score_models import ScoreModule
    from fourierdiffusion_amd.schedulers.sde import VPScheduler
    from fourierdiffusion_amd.utils.dataclasses import DiffusableBatch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = ScoreModule(n_channels=3, max_len=20, noise_scheduler=VPScheduler(), d_model=8, num_layers=1, n_head=4)
    with pytest.raises(_C.FdError):
        m(DiffusableBatch(X=torch.zeros(2, 20, 3), timesteps=torch.ones(2)))
    from fourierdiffusion_amd.utils.fourier import dft
    with pytest.raises(_C.FdError):
        dft(torch.zeros(2, 20, 3))
