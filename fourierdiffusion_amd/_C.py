"""ctypes binding of libfdiff_hip.so (include/fdiff_hip.h).

torch tensors are used only as device-memory containers: every call passes
``tensor.data_ptr()`` and the current HIP stream handle.  There is NO CPU
fallback -- if the shared library is missing, or a tensor is not on a GPU,
the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FDIFF_LIB", os.path.join(_HERE, "libfdiff_hip.so"))   # FDIFF_LIB: kernel-ablation builds

FD_MODE_F32 = 0
FD_MODE_BF16 = 1
FD_COMM_ID_BYTES = 128
FD_BACKBONE_TRANSFORMER, FD_BACKBONE_MLP, FD_BACKBONE_LSTM = 0, 1, 2


class FdError(RuntimeError):
    pass


class SdeParams(C.Structure):
    _fields_ = [("kind", C.c_int), ("p0", C.c_float), ("p1", C.c_float)]


class ModelDims(C.Structure):
    _fields_ = [("n_channels", C.c_int), ("max_len", C.c_int), ("d_model", C.c_int),
                ("n_head", C.c_int), ("num_layers", C.c_int), ("dim_ff", C.c_int)]


class ParamEntry(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("offset", C.c_int64), ("numel", C.c_int64),
                ("rows", C.c_int32), ("cols", C.c_int32), ("trainable", C.c_int32)]


_vp = C.c_void_p
_PROTOS = {
    "fd_version": (C.c_int, []),
    "fd_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "fd_ctx_destroy": (C.c_int, [_vp]),
    "fd_last_error": (C.c_char_p, [_vp]),
    "fd_ctx_workspace_bytes": (C.c_size_t, [_vp]),
    "fd_ctx_check": (C.c_int, [_vp]),
    "fd_ctx_rearm": (C.c_int, [_vp]),
    "fd_prof_shader_clock_mhz": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "fd_mega_jit_compile": (C.c_int, [C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "fd_prof_begin": (C.c_int, [_vp]),
    "fd_prof_stride": (C.c_int, [_vp, C.c_int]),
    "fd_prof_end": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "fd_rfft_pack": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fd_irfft_unpack": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fd_rfft_pack_standardize": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fd_destandardize_irfft": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fd_spectral_density": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fd_localization_metrics": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fd_frequency_smooth": (C.c_int, [_vp, _vp, C.c_float, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fd_randn": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint64, C.c_uint64, _vp]),
    "fd_philox_words": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint64, C.c_uint64, _vp]),
    "fd_dropout_decisions": (C.c_int, [_vp, _vp, C.c_size_t, C.c_float, C.c_uint64, C.c_uint64, _vp]),
    "fd_prior_sample": (C.c_int, [_vp, C.POINTER(SdeParams), _vp, _vp, C.c_uint64, C.c_uint64, _vp,
                                  C.c_int, C.c_int, C.c_int, _vp]),
    "fd_sde_step": (C.c_int, [_vp, C.POINTER(SdeParams), _vp, _vp, _vp, _vp, C.c_uint64, C.c_uint64,
                              C.c_double, C.c_float, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fd_perturb": (C.c_int, [_vp, C.POINTER(SdeParams), _vp, _vp, _vp, _vp, C.c_uint64, C.c_uint64,
                             _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fd_dsm_loss": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fd_positional_add": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_float, _vp]),
    "fd_time_embed_add": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fd_score_param_count": (C.c_int64, [C.POINTER(ModelDims)]),
    "fd_score_layout": (C.c_int, [C.POINTER(ModelDims), C.POINTER(ParamEntry), C.POINTER(C.c_int)]),
    "fd_score_create": (C.c_int, [_vp, C.POINTER(ModelDims), C.POINTER(_vp)]),
    "fd_score_param_count_ex": (C.c_int64, [C.POINTER(ModelDims), C.c_int, C.c_int]),
    "fd_score_layout_ex": (C.c_int, [C.POINTER(ModelDims), C.c_int, C.c_int, C.POINTER(ParamEntry), C.POINTER(C.c_int)]),
    "fd_score_create_ex": (C.c_int, [_vp, C.POINTER(ModelDims), C.c_int, C.c_int, C.POINTER(_vp)]),
    "fd_score_destroy": (C.c_int, [_vp]),
    "fd_score_prepare": (C.c_int, [_vp, _vp, _vp]),
    "fd_score_forward": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "fd_score_plan": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int)]),
    "fd_score_set_train_mode": (C.c_int, [_vp, C.c_int]),
    "fd_score_forward_train": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_float, C.c_uint64, C.c_uint64, _vp]),
    "fd_score_backward": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "fd_score_train_plan": (C.c_int, [_vp, C.c_int, C.c_char_p, C.POINTER(C.c_int)]),
    "fd_score_train_dsm_supported": (C.c_int, [_vp, C.c_int]),
    "fd_score_train_dsm": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_float, C.c_int, C.c_float, C.c_uint64, C.c_uint64,
                                     _vp, _vp, C.c_int, _vp]),
    "fd_sampler_run": (C.c_int, [_vp, C.POINTER(SdeParams), _vp, _vp, C.c_int, C.c_float, _vp, _vp,
                                 C.c_uint64, C.c_uint64, C.c_int, C.c_int, _vp]),
    "fd_langevin_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint64, C.c_uint64, C.c_float, C.c_float, _vp, C.c_int, C.c_int,
                                   C.c_int, _vp]),
    "fd_sampler_run_pc": (C.c_int, [_vp, C.POINTER(SdeParams), _vp, _vp, C.c_int, C.c_float, _vp, _vp, _vp, C.c_int, C.c_float,
                                    C.c_uint64, C.c_uint64, C.c_int, C.c_int, _vp]),
    "fd_grad_sqnorm": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp]),
    "fd_adamw_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_float, C.c_float,
                                C.c_float, C.c_float, C.c_float, _vp, C.c_float, C.c_float, C.c_int64,
                                C.c_int64, _vp]),
    "fd_comm_unique_id": (C.c_int, [_vp]),
    "fd_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "fd_comm_destroy": (C.c_int, [_vp]),
    "fd_comm_rccl_path": (C.c_int, [C.c_char_p, C.c_int]),
    "fd_allreduce_grads": (C.c_int, [_vp, _vp, C.c_int64, C.c_float, _vp]),
    "fd_project_rows": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "fd_transpose_rows": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "fd_sort_rows_temp_bytes": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "fd_sort_rows": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
    "fd_w2_sorted_rows": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libfdiff_hip.so once; raise (never fall back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FdError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                "(make -C fourierdiffusion_amd/csrc). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


# ---------------------------------------------------------------- contexts
_ctxs: dict = {}


def ctx(device: torch.device) -> int:
    """One fd_ctx per (process, device)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise FdError(f"the HIP engine needs GPU tensors, got device '{device}' (no CPU fallback)")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _ctxs:
        h = _vp()
        rc = lib().fd_ctx_create(idx, C.byref(h))
        if rc != 0:
            raise FdError(f"fd_ctx_create(device={idx}) failed with code {rc}")
        _ctxs[idx] = h
    return _ctxs[idx]


def check(rc: int, ctx_handle) -> None:
    if rc != 0:
        msg = lib().fd_last_error(ctx_handle)
        raise FdError(f"libfdiff_hip error {rc}: {msg.decode() if msg else '?'}")


def stream_of(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def dev_f32(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    """Validate a tensor as an engine input: GPU, float32, contiguous."""
    if not isinstance(t, torch.Tensor):
        raise FdError(f"{name} must be a torch.Tensor")
    if t.device.type != "cuda":
        raise FdError(f"{name} is on '{t.device}': the HIP engine needs GPU tensors (no CPU fallback)")
    if t.dtype != torch.float32:
        t = t.float()
    if not t.is_contiguous():
        t = t.contiguous()
    return t


def model_dims(n_channels, max_len, d_model, n_head, num_layers, dim_ff=2048) -> ModelDims:
    return ModelDims(int(n_channels), int(max_len), int(d_model), int(n_head), int(num_layers), int(dim_ff))


def score_layout(dims: ModelDims, backbone: int = 0, d_mlp: int = 0):
    """[(name, offset, numel, shape, trainable)] + total float count, straight from the engine."""
    n = C.c_int(0)
    rc = lib().fd_score_layout_ex(C.byref(dims), backbone, d_mlp, None, C.byref(n))
    if rc != 0:
        raise FdError(f"fd_score_layout failed ({rc}): bad model dims")
    arr = (ParamEntry * n.value)()
    rc = lib().fd_score_layout_ex(C.byref(dims), backbone, d_mlp, arr, C.byref(n))
    if rc != 0:
        raise FdError(f"fd_score_layout failed ({rc})")
    out = []
    for e in arr:
        shape = (e.rows, e.cols) if e.cols else (e.rows,)
        out.append((e.name.decode(), int(e.offset), int(e.numel), shape, bool(e.trainable)))
    total = int(lib().fd_score_param_count_ex(C.byref(dims), backbone, d_mlp))
    return out, total
