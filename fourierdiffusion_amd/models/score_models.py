"""ScoreModule on the HIP engine -- same surface as fdiff.models.score_models.ScoreModule
(reference: src/fdiff/models/score_models.py:22-166).

The network (Linear embed + learned positional table with max_norm + Gaussian-Fourier time
embedding -> L x post-LN transformer encoder layer (relu, dim_ff 2048) -> Linear unembed) lives in
ONE flat fp32 device buffer whose layout is defined by the engine (``fd_score_layout``); the
reference's state_dict keys are views into it, so checkpoints interchange, the optimizer is one
fused pass and the data-parallel gradient exchange is one flat RCCL all-reduce.

No autograd graph is built: ``forward`` in training mode keeps activations inside the engine and
``backward(dscore)`` accumulates into ``self.grads`` (same flat layout).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from collections import OrderedDict
from typing import Any, Callable, Dict, Iterator, Optional, Tuple

import torch

from .. import _C, _rng
from ..schedulers.sde import SDE
from ..utils.dataclasses import DiffusableBatch
from ..utils.losses import get_sde_loss_fn

_PRECISIONS = {"fp32": _C.FD_MODE_F32, "bf16": _C.FD_MODE_BF16}


def _default_precision() -> str:
    return os.environ.get("FDIFF_PRECISION", "bf16")


class ScoreModule:
    # torch.nn.TransformerEncoderLayer defaults the reference relies on (score_models.py:57-59)
    dim_feedforward = 2048
    dropout = 0.1
    _backbone = _C.FD_BACKBONE_TRANSFORMER
    _d_mlp = 0

    def __init__(
        self,
        n_channels: int,
        max_len: int,
        noise_scheduler: SDE,
        fourier_noise_scaling: bool = True,
        d_model: int = 60,
        num_layers: int = 3,
        n_head: int = 12,
        num_training_steps: int = 1000,
        lr_max: float = 1e-3,
        likelihood_weighting: bool = False,
    ) -> None:
        self.max_len = int(max_len)
        self.n_channels = int(n_channels)
        self.noise_scheduler = noise_scheduler
        self.num_warmup_steps = num_training_steps // 10
        self.num_training_steps = num_training_steps
        self.lr_max = lr_max
        self.d_model = int(d_model)
        self.num_layers = int(num_layers)
        self.n_head = int(n_head)
        self.scale_noise = fourier_noise_scaling
        self.likelihood_weighting = likelihood_weighting
        self.training = True
        self.precision = _default_precision()      # eval/sampling arithmetic: "bf16" (MFMA) or "fp32" (parity)
        # training arithmetic (forward with dropout + backward): "bf16" = fused MFMA kernels with fp32 accumulation where they
        # are instantiated for the model's dims (else the exact-f32 kernels), "fp32" = exact-f32 kernels (parity anchor)
        self.train_precision = os.environ.get("FDIFF_TRAIN_PRECISION", "bf16")
        self._train_mode_set: Optional[str] = None
        self.hparams: Dict[str, Any] = dict(
            n_channels=n_channels, max_len=max_len, noise_scheduler=noise_scheduler,
            fourier_noise_scaling=fourier_noise_scaling, d_model=d_model, num_layers=num_layers, n_head=n_head,
            num_training_steps=num_training_steps, lr_max=lr_max, likelihood_weighting=likelihood_weighting)

        self.training_loss_fn, self.validation_loss_fn = self.set_loss_fn()

        if self.d_model % self.n_head != 0:
            raise AssertionError("embed_dim must be divisible by num_heads")
        self._dims = _C.model_dims(self.n_channels, self.max_len, self.d_model, self.n_head, self.num_layers,
                                   self.dim_feedforward)
        self._layout, self._nparams = _C.score_layout(self._dims, self._backbone, self._d_mlp)
        self._flat = torch.zeros(self._nparams, dtype=torch.float32)
        self._grads: Optional[torch.Tensor] = None
        self._zero_pending = False
        self._views: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._trainable: Dict[str, bool] = {}
        self._bind_views()
        self._init_parameters()
        self._handle: Optional[int] = None
        self._dirty = True

    # ------------------------------------------------------------------ parameters
    def _bind_views(self) -> None:
        self._views.clear()
        for name, off, numel, shape, trainable in self._layout:
            self._views[name] = self._flat[off:off + numel].view(*shape)
            self._trainable[name] = trainable

    def _init_parameters(self) -> None:
        """Same initialisers, drawn from torch's global generator in the same order as the reference's
        constructor (score_models.py:52-62), so that ``torch.manual_seed(s)`` gives the reference's weights:
        Embedding N(0,1); randn*30; Linear = kaiming_uniform(a=sqrt 5) + U(+-1/sqrt(fan_in)) bias;
        MHA: out_proj Linear init, then xavier_uniform in_proj, zero in_proj_bias / out_proj.bias;
        one encoder layer is drawn and copied into all L layers (nn.TransformerEncoder deep-copies)."""
        v = self._views
        init = torch.nn.init

        def linear(prefix: str) -> None:
            w, b = v[prefix + ".weight"], v[prefix + ".bias"]
            init.kaiming_uniform_(w, a=math.sqrt(5))
            bound = 1.0 / math.sqrt(w.shape[1]) if w.shape[1] > 0 else 0.0
            init.uniform_(b, -bound, bound)

        if "pos_encoder.embedding.weight" in v:
            init.normal_(v["pos_encoder.embedding.weight"])
        v["time_encoder.W"].copy_(torch.randn((self.d_model + 1) // 2) * 30.0)
        linear("time_encoder.dense")
        linear("embedder")
        linear("unembedder")
        self._init_backbone(v, linear)

    def _init_backbone(self, v, linear) -> None:
        init = torch.nn.init
        if self.num_layers > 0:
            p = "backbone.layers.0."
            linear(p + "self_attn.out_proj")
            init.xavier_uniform_(v[p + "self_attn.in_proj_weight"])
            v[p + "self_attn.in_proj_bias"].zero_()
            v[p + "self_attn.out_proj.bias"].zero_()
            linear(p + "linear1")
            linear(p + "linear2")
            for nm in ("norm1", "norm2"):
                v[p + nm + ".weight"].fill_(1.0)
                v[p + nm + ".bias"].zero_()
            for i in range(1, self.num_layers):
                q = f"backbone.layers.{i}."
                for k in list(v):
                    if k.startswith(p):
                        v[q + k[len(p):]].copy_(v[k])

    def parameters(self) -> Iterator[torch.Tensor]:
        return iter(self._views.values())

    def named_parameters(self) -> Iterator[Tuple[str, torch.Tensor]]:
        return iter(self._views.items())

    def trainable_mask(self) -> Dict[str, bool]:
        return dict(self._trainable)

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, t.detach().clone()) for k, t in self._views.items())

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True) -> None:
        missing = [k for k in self._views if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._views]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
        for k, dst in self._views.items():
            if k in state_dict:
                src = state_dict[k]
                if tuple(src.shape) != tuple(dst.shape):
                    raise RuntimeError(f"size mismatch for {k}: {tuple(src.shape)} vs {tuple(dst.shape)}")
                dst.copy_(src.to(device=dst.device, dtype=torch.float32))
        self._dirty = True

    @property
    def flat_parameters(self) -> torch.Tensor:
        return self._flat

    def mark_parameters_changed(self) -> None:
        """Call after writing into ``flat_parameters`` / the views (e.g. an optimizer step)."""
        self._dirty = True

    # ------------------------------------------------------------------ device / mode
    @property
    def device(self) -> torch.device:
        return self._flat.device

    def to(self, device=None, **kwargs) -> "ScoreModule":
        device = torch.device(device if device is not None else kwargs.get("device"))
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device != self._flat.device:
            self._release()
            self._flat = self._flat.to(device)
            self.grads = None
            self._bind_views()
            self._dirty = True
        return self

    def cuda(self) -> "ScoreModule":
        return self.to("cuda")

    def cpu(self) -> "ScoreModule":
        return self.to("cpu")

    def eval(self) -> "ScoreModule":
        self.training = False
        return self

    def train(self, mode: bool = True) -> "ScoreModule":
        self.training = bool(mode)
        return self

    def _release(self) -> None:
        if self._handle is not None:
            _C.lib().fd_score_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _engine(self) -> Tuple[int, int]:
        """(ctx, model handle); creates the engine object and refreshes derived weights when dirty."""
        if self._flat.device.type != "cuda":
            raise _C.FdError("ScoreModule is on the CPU: move it to the GPU with .to('cuda') -- the score network "
                             "runs only on the HIP engine (no CPU fallback)")
        ctx = _C.ctx(self._flat.device)
        if self._handle is None:
            h = C.c_void_p()
            rc = _C.lib().fd_score_create_ex(ctx, C.byref(self._dims), self._backbone, self._d_mlp, C.byref(h))
            _C.check(rc, ctx)
            self._handle = h
            self._dirty = True
            self._train_mode_set = None
        if self._dirty:
            rc = _C.lib().fd_score_prepare(self._handle, self._flat.data_ptr(), _C.stream_of(self._flat))
            _C.check(rc, ctx)
            self._dirty = False
        if self._train_mode_set != self.train_precision:
            rc = _C.lib().fd_score_set_train_mode(self._handle, _PRECISIONS[self.train_precision])
            if rc == -5 and self.train_precision == "bf16":       # FD_ERR_UNSUPPORTED: dims without bf16 training kernels
                rc = _C.lib().fd_score_set_train_mode(self._handle, _C.FD_MODE_F32)
                self.train_mode_effective = "fp32"
            else:
                self.train_mode_effective = self.train_precision
            _C.check(rc, ctx)
            self._train_mode_set = self.train_precision
        return ctx, self._handle

    @property
    def precision_effective(self) -> str:
        """The arithmetic eval / sampling really runs in: ``precision`` unless it asks for bf16 and this model's dimensions
        have no bf16 MFMA instantiation (fd_score_plan answers FD_ERR_UNSUPPORTED) -- then the exact-f32 kernels run, the same
        way ``train_mode_effective`` reports the training fallback.  The reference takes any d_model / n_head
        (score_models.py:23-65); a width the MFMA path does not cover must still sample."""
        if self.precision != "bf16":
            return self.precision
        ctx, h = self._engine()
        key = (int(h.value or 0), self.precision)
        if getattr(self, "_precision_probe", (None, None))[0] != key:
            buf = C.create_string_buffer(192)
            rc = _C.lib().fd_score_plan(h, 1, _C.FD_MODE_BF16, buf, None)
            if rc not in (0, -5):
                _C.check(rc, ctx)
            self._precision_probe = (key, "bf16" if rc == 0 else "fp32")
        return self._precision_probe[1]

    def plan(self, batch_size: int, precision: Optional[str] = None) -> Tuple[str, int]:
        """(description of the kernel path a batch of this size takes, series per workgroup) -- fd_score_plan."""
        ctx, h = self._engine()
        buf = C.create_string_buffer(192)
        spw = C.c_int(0)
        rc = _C.lib().fd_score_plan(h, int(batch_size), _PRECISIONS[precision or self.precision_effective], buf, C.byref(spw))
        _C.check(rc, ctx)
        return buf.value.decode(), spw.value

    def train_plan(self, batch_size: int) -> Tuple[str, int]:
        """(description of the training launch plan of a batch of this size, token splits of the weight-gradient kernel; 0 on
        the exact-f32 path) -- fd_score_train_plan."""
        ctx, h = self._engine()
        buf = C.create_string_buffer(192)
        ts = C.c_int(0)
        _C.check(_C.lib().fd_score_train_plan(h, int(batch_size), buf, C.byref(ts)), ctx)
        return buf.value.decode(), ts.value

    # ------------------------------------------------------------------ forward / backward
    def forward(self, batch: DiffusableBatch) -> torch.Tensor:
        X = batch.X
        assert X.size()[1:] == (self.max_len, self.n_channels), (
            f"X has wrong shape, should be {(X.size(0), self.max_len, self.n_channels)}, but is {X.size()}")
        timesteps = batch.timesteps
        assert timesteps is not None and timesteps.size(0) == len(batch)
        ctx, h = self._engine()
        Xd = _C.dev_f32(X.to(self.device), "batch.X")
        td = _C.dev_f32(timesteps.to(self.device), "batch.timesteps")
        out = torch.empty_like(Xd)
        B = Xd.shape[0]
        if self.training:
            p = float(self.dropout)
            key, off = _rng.stream()
            rc = _C.lib().fd_score_forward_train(h, Xd.data_ptr(), td.data_ptr(), out.data_ptr(), B, p,
                                                 key, off, _C.stream_of(Xd))
            self._train_inputs = (Xd, td)       # the engine reads them again in backward
        else:
            mode = _PRECISIONS[self.precision_effective]
            rc = _C.lib().fd_score_forward(h, Xd.data_ptr(), td.data_ptr(), out.data_ptr(), B, mode,
                                           _C.stream_of(Xd))
        _C.check(rc, ctx)
        return out

    __call__ = forward

    def train_dsm(self, x_noisy: torch.Tensor, timesteps: torch.Tensor, target: torch.Tensor, std: torch.Tensor,
                  likelihood_weighting: bool = False, grad_weight: float = 1.0) -> Optional[torch.Tensor]:
        """Training forward + denoising score-matching loss + backward as one engine call (fd_score_train_dsm): the loss
        tensor, with the gradients ACCUMULATED into ``self.grads`` -- or None when this model has no fused step (exact-f32
        training, MLP / LSTM backbones, very wide C * d_model), in which case the caller runs forward -> fd_dsm_loss ->
        backward.  Same Philox stream use as ``forward`` in training mode (one key per call)."""
        if not self.training or getattr(self, "_no_fused_dsm", False):     # (_no_fused_dsm: a manual switch, never latched here)
            return None
        ctx, h = self._engine()
        if self.train_mode_effective != "bf16":
            return None
        # capability per batch size, asked BEFORE a Philox key is drawn (an unsupported batch must not consume one: the trainer
        # counts keys per step to keep empty-slice ranks in step) and cached per B -- one oversized batch does not switch the
        # fused step off for later, smaller ones
        # (key: the engine handle the answer was given for, the batch size, and the one environment switch the C side reads per
        # call -- a recreated handle or a toggled FDIFF_TRAIN_DSM_UNFUSED must not meet a stale verdict)
        Bn = int(x_noisy.shape[0])
        cache = self.__dict__.setdefault("_fused_dsm_ok", {})
        ckey = (int(h.value) if hasattr(h, "value") else int(h), Bn, os.environ.get("FDIFF_TRAIN_DSM_UNFUSED"))
        ok = cache.get(ckey)
        if ok is None:
            ok = cache[ckey] = bool(_C.lib().fd_score_train_dsm_supported(h, Bn))
        if not ok:
            return None
        Xd = _C.dev_f32(x_noisy.to(self.device), "x_noisy")
        td = _C.dev_f32(timesteps.to(self.device), "timesteps")
        tg = _C.dev_f32(target, "target")
        sd = _C.dev_f32(std, "std")
        grads, acc = self._grads_for_backward()
        loss = torch.empty(1, device=self.device, dtype=torch.float32)
        key, off = _rng.stream()
        rc = _C.lib().fd_score_train_dsm(h, Xd.data_ptr(), td.data_ptr(), tg.data_ptr(), sd.data_ptr(),
                                         1 if likelihood_weighting else 0, float(grad_weight), Xd.shape[0], float(self.dropout),
                                         key, off, loss.data_ptr(), grads.data_ptr(), acc, _C.stream_of(Xd))
        _C.check(rc, ctx)
        self._train_inputs = (Xd, td, tg, sd)
        return loss[0]

    @property
    def grads(self) -> Optional[torch.Tensor]:
        """The flat gradient buffer (same layout as ``flat_parameters``).  ``zero_grad()`` is lazy: the next backward OVERWRITES the
        buffer (the engine's accumulate = 0 form, bit-identical to accumulating onto zeros: tests/test_gpu_train_bf16.py) instead of
        a 13 MB fill kernel in front of every step plus a read of it in every reduce; anyone who looks at the buffer before that
        backward gets the zeros here."""
        if self._zero_pending and self._grads is not None:
            self._grads.zero_()
        self._zero_pending = False
        return self._grads

    @grads.setter
    def grads(self, value: Optional[torch.Tensor]) -> None:
        self._grads = value
        self._zero_pending = False

    def _grads_for_backward(self) -> Tuple[torch.Tensor, int]:
        """(buffer, accumulate flag of the engine call) -- consumes a pending ``zero_grad()``."""
        if self._grads is None or self._grads.device != self.device:
            self._grads = torch.zeros_like(self._flat)
            self._zero_pending = False
        acc = 0 if self._zero_pending else 1
        self._zero_pending = False
        return self._grads, acc

    def zero_grad(self) -> None:
        if self._grads is not None:
            if os.environ.get("FDIFF_LAZY_ZERO_GRAD", "1") == "0":      # (A/B runs: the fill kernel of rounds 1-5)
                self._grads.zero_()
                self._zero_pending = False
            else:
                self._zero_pending = True

    def backward(self, dscore: torch.Tensor, accumulate: bool = True) -> torch.Tensor:
        """d loss / d params for the last training-mode forward; accumulates into ``self.grads``."""
        ctx, h = self._engine()
        grads, acc = self._grads_for_backward()
        d = _C.dev_f32(dscore, "dscore")
        rc = _C.lib().fd_score_backward(h, d.data_ptr(), grads.data_ptr(), acc if accumulate else 0, _C.stream_of(d))
        _C.check(rc, ctx)
        return grads

    def grad_views(self) -> "OrderedDict[str, torch.Tensor]":
        assert self.grads is not None
        return OrderedDict((name, self.grads[off:off + numel].view(*shape))
                           for name, off, numel, shape, _ in self._layout)

    # ------------------------------------------------------------------ Lightning-style hooks
    def training_step(self, batch: DiffusableBatch, batch_idx: int = 0, dataloader_idx: int = 0,
                      grad_weight: float = 1.0) -> torch.Tensor:
        return self.training_loss_fn(self, batch, grad_weight=grad_weight)

    def validation_step(self, batch: DiffusableBatch, batch_idx: int = 0, dataloader_idx: int = 0) -> torch.Tensor:
        return self.validation_loss_fn(self, batch)

    def configure_optimizers(self):
        """AdamW(lr_max) + cosine schedule with warmup = num_training_steps // 10, stepped per batch
        (score_models.py:122-130)."""
        from ..optim import FusedAdamW, cosine_schedule_with_warmup
        opt = FusedAdamW(self, lr=self.lr_max)
        sched = cosine_schedule_with_warmup(self.num_warmup_steps, self.num_training_steps)
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sched, "interval": "step"}}

    def set_loss_fn(self) -> Tuple[Callable, Callable]:
        if isinstance(self.noise_scheduler, SDE):
            return (get_sde_loss_fn(self.noise_scheduler, train=True, likelihood_weighting=self.likelihood_weighting),
                    get_sde_loss_fn(self.noise_scheduler, train=False, likelihood_weighting=self.likelihood_weighting))
        raise NotImplementedError(
            f"Scheduler {self.noise_scheduler} not implemented yet, cannot set loss function.")

    # ------------------------------------------------------------------ checkpoints (Lightning layout)
    def save_checkpoint(self, path, **extra) -> None:
        # Lightning's layout (state_dict keys of the reference model, ctor hyper-parameters incl. the scheduler object) plus
        # the bookkeeping keys its load_from_checkpoint expects, so the reference can load checkpoints written here
        ckpt = {"epoch": 0, "global_step": 0, "pytorch-lightning_version": "2.1.0",
                "state_dict": OrderedDict((k, t.cpu()) for k, t in self.state_dict().items()),
                "hparams_name": "kwargs", "hyper_parameters": dict(self.hparams), "engine": "fourierdiffusion_amd"}
        ckpt.update(extra)
        torch.save(ckpt, path)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, **overrides) -> "ScoreModule":
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        hp = dict(ckpt["hyper_parameters"])
        hp.update(overrides)
        model = cls(**hp)
        model.load_state_dict(ckpt["state_dict"])
        if map_location is not None:
            model.to(map_location)
        return model


class MLPScoreModule(ScoreModule):
    """fdiff.models.score_models.MLPScoreModule (score_models.py:169-246) on the engine: the series is flattened to
    (B, T*C), Linear embed, + time embedding, num_layers x { h += torchvision.ops.MLP(d_model, [d_mlp, d_model], dropout 0.1)(h) },
    Linear unembed.  Exact-f32 kernels (csrc/fd_backbones.hip); state-dict keys backbone.{i}.0|3.weight|bias."""
    _backbone = _C.FD_BACKBONE_MLP

    def __init__(self, n_channels: int, max_len: int, noise_scheduler: SDE, fourier_noise_scaling: bool = True, d_model: int = 72,
                 d_mlp: int = 512, num_layers: int = 3, num_training_steps: int = 1000, lr_max: float = 1e-3,
                 likelihood_weighting: bool = False) -> None:
        self._d_mlp = int(d_mlp)
        self.d_mlp = int(d_mlp)
        super().__init__(n_channels=n_channels, max_len=max_len, noise_scheduler=noise_scheduler,
                         fourier_noise_scaling=fourier_noise_scaling, d_model=d_model, num_layers=num_layers, n_head=1,
                         num_training_steps=num_training_steps, lr_max=lr_max, likelihood_weighting=likelihood_weighting)
        self.hparams = dict(n_channels=n_channels, max_len=max_len, noise_scheduler=noise_scheduler,
                            fourier_noise_scaling=fourier_noise_scaling, d_model=d_model, d_mlp=d_mlp, num_layers=num_layers,
                            num_training_steps=num_training_steps, lr_max=lr_max, likelihood_weighting=likelihood_weighting)
        self.precision = "fp32"
        self.train_precision = "fp32"

    def _init_backbone(self, v, linear) -> None:
        # nn.Linear default initialisation of every layer, drawn independently per block (a ModuleList of fresh MLPs)
        for i in range(self.num_layers):
            linear(f"backbone.{i}.0")
            linear(f"backbone.{i}.3")


class LSTMScoreModule(ScoreModule):
    """fdiff.models.score_models.LSTMScoreModule (score_models.py:249-317) on the engine: Linear embed, + time embedding,
    num_layers x { h += nn.LSTM(d_model, d_model, batch_first=True)(h)[0] }, Linear unembed.  Exact-f32 kernels; state-dict keys
    backbone.{i}.weight_ih_l0 | weight_hh_l0 | bias_ih_l0 | bias_hh_l0 (gate order i, f, g, o)."""
    _backbone = _C.FD_BACKBONE_LSTM

    def __init__(self, n_channels: int, max_len: int, noise_scheduler: SDE, fourier_noise_scaling: bool = True, d_model: int = 72,
                 num_layers: int = 3, num_training_steps: int = 1000, lr_max: float = 1e-3,
                 likelihood_weighting: bool = False) -> None:
        super().__init__(n_channels=n_channels, max_len=max_len, noise_scheduler=noise_scheduler,
                         fourier_noise_scaling=fourier_noise_scaling, d_model=d_model, num_layers=num_layers, n_head=1,
                         num_training_steps=num_training_steps, lr_max=lr_max, likelihood_weighting=likelihood_weighting)
        self.hparams = dict(n_channels=n_channels, max_len=max_len, noise_scheduler=noise_scheduler,
                            fourier_noise_scaling=fourier_noise_scaling, d_model=d_model, num_layers=num_layers,
                            num_training_steps=num_training_steps, lr_max=lr_max, likelihood_weighting=likelihood_weighting)
        self.precision = "fp32"
        self.train_precision = "fp32"

    def _init_backbone(self, v, linear) -> None:
        # nn.LSTM.reset_parameters: every tensor U(-1/sqrt(hidden), 1/sqrt(hidden))
        k = 1.0 / math.sqrt(self.d_model)
        for i in range(self.num_layers):
            for nm in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
                torch.nn.init.uniform_(v[f"backbone.{i}.{nm}"], -k, k)
