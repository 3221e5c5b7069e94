"""Drop-in DiT3D backbone (the reference's Kinetics-600 model) backed by libdfot_hip.so.

Mirrors the reference's plugin contract for this path:
  * constructor keywords of DiscreteDiffusion._build_model (algorithms/dfot/diffusion/discrete_diffusion.py:64-92)
    and DiT3D.__init__ (algorithms/dfot/backbones/dit/dit3d.py:13-83): variant "full", pos_emb_type "rope_3d",
    no external condition, causal masking rejected exactly as the reference does;
  * ``forward(x, noise_levels, external_cond=None, external_cond_mask=None)`` (dit3d.py:146-192) with integer
    ``noise_levels`` -- the level index DiscreteDiffusion.model_predictions passes (discrete_diffusion.py:173-174);
  * state-dict key names / shapes / order of the reference module, so its checkpoints load with ``load_state_dict``.
Parameters live here as fp32 ``nn.Parameter``s; the C library keeps packed bf16 copies and a per-level modulation table
that are rebuilt whenever a parameter changes.  Inference only.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import nn

from . import capi, ops
from .backbone import _Node, _get


class DiT3D(nn.Module):
    def __init__(self, cfg, x_shape: Sequence[int], max_tokens: int, external_cond_type: str = "action",
                 external_cond_num_classes: Optional[int] = None, external_cond_dim: int = 0,
                 use_causal_mask: bool = False, timesteps: int = 1000, **kwargs):
        if use_causal_mask:
            raise NotImplementedError("Causal masking is not yet implemented for DiT3D backbone")
        super().__init__()
        if external_cond_dim:
            raise ValueError("external conditions are not supported by the DiT3D engine (kinetics_600 has none)")
        self.cfg = cfg
        self.x_shape = tuple(int(v) for v in x_shape)
        self.external_cond_dim = 0
        self.use_causal_mask = False
        self.patch_size = int(_get(cfg, "patch_size", 2))
        c = capi.DiTConfig()
        c.depth = int(_get(cfg, "depth"))
        c.num_heads = int(_get(cfg, "num_heads"))
        c.patch_size = self.patch_size
        c.in_channels, c.height, c.width = self.x_shape
        c.noise_dim = 256
        c.timesteps = int(timesteps)
        c.rope_theta = 10000.0
        c.eps = 1e-6
        self._configure(c, cfg, int(max_tokens))
        self.hidden_size = int(c.hidden_size)
        self.max_tokens = int(c.max_tokens)
        self._ccfg = c
        self.num_patches = (c.height // c.patch_size) * (c.width // c.patch_size)
        self._handle = C.c_void_p()
        capi.check(capi.lib.dfot_dit_create(C.byref(c), C.byref(self._handle)))
        self._names = []
        shape = (C.c_int64 * 4)()
        ndim = C.c_int()
        for i in range(capi.lib.dfot_dit_num_params(self._handle)):
            name = capi.lib.dfot_dit_param_name(self._handle, i).decode()
            capi.check(capi.lib.dfot_dit_param_shape(self._handle, i, shape, C.byref(ndim)))
            self._register(name, tuple(shape[k] for k in range(ndim.value)))
            self._names.append(name)
        self._synced: Optional[Tuple] = None
        self._reserved = 0
        self._op_key: Optional[int] = None
        # training form (autograd): the saved-activation engine of trainer.DiT3DTrainer, built at the first forward under grad
        self._ctor = dict(max_tokens=int(max_tokens), timesteps=int(timesteps))
        self._trainer = None
        self._trainer_sig = None
        self._train_stamp = 0  # counts training forwards (see backbone.UViT3DPose._train_backward_impl)
        self._train_names = [n for n, _ in self.named_parameters()]

    def _configure(self, c: "capi.DiTConfig", cfg, max_tokens: int) -> None:
        """dit3d.yaml keys -> engine config (variant 0)."""
        if _get(cfg, "variant", "full") != "full":
            raise ValueError(f"unsupported DiT variant {_get(cfg, 'variant')!r}: DiT3D builds 'full' (see DifferenceDiT3D)")
        if _get(cfg, "pos_emb_type", "rope_3d") != "rope_3d":
            raise ValueError("only pos_emb_type='rope_3d' is supported")
        ratio = _get(cfg, "spatial_mlp_ratio", None)
        c.hidden_size = int(_get(cfg, "hidden_size"))
        c.max_tokens = max_tokens
        c.mlp_hidden = int(c.hidden_size * ratio) if ratio else 0
        c.variant = 0

    @property
    def in_channels(self) -> int:
        return self.x_shape[0]

    @property
    def noise_level_dim(self) -> int:
        return 256

    def _register(self, name: str, shape: Tuple[int, ...]) -> None:
        *path, leaf = name.split(".")
        node: nn.Module = self
        for part in path:
            if part not in node._modules:
                node.add_module(part, _Node())
            node = node._modules[part]
        node.register_parameter(leaf, nn.Parameter(torch.zeros(shape, dtype=torch.float32)))

    def _tensors(self) -> Dict[str, torch.Tensor]:
        return dict(self.named_parameters())

    def reset_parameters(self, seed: int = 0) -> None:
        """The reference's init (dit3d.py:92-109, dit_blocks.py:392-395,422-425,476-486,528-531): xavier-uniform Linear
        weights, N(0, 0.02) embedding MLP, zero biases, zero modulations and zero final projection."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, t in self._tensors().items():
                if name.endswith(("bias", "qkv_bias", "proj_bias")) or ".modulation." in name or name.startswith("dit_base.final_layer.linear"):
                    t.zero_()
                elif name.startswith(("noise_level_pos_embedding", "diff_embedder")):
                    t.copy_(0.02 * torch.randn(t.shape, generator=g))
                else:
                    fan_out, fan_in = t.shape[0], math.prod(t.shape[1:])
                    bound = math.sqrt(6.0 / (fan_in + fan_out))
                    t.copy_((torch.rand(t.shape, generator=g) * 2 - 1) * bound)

    def init_random(self, seed: int = 0) -> None:
        """Non-degenerate random weights for benchmarks (the reference zero-inits every modulation and the final
        projection, which makes the output identically zero): weights ~ N(0, 1/fan_in) (modulations at half gain),
        biases ~ N(0, 0.05^2) -- the distribution of oracle.dit.seeded_params."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, t in self._tensors().items():
                if name.endswith(".bias"):
                    v = 0.05 * torch.randn(t.shape, generator=g)
                else:
                    gain = 0.5 if ".modulation." in name else 1.0
                    v = gain * torch.randn(t.shape, generator=g) / math.sqrt(math.prod(t.shape[1:]))
                t.copy_(v.to(t.device))

    def _signature(self) -> Tuple:
        return tuple((t.data_ptr(), t._version) for t in self._tensors().values())

    def sync_weights(self, force: bool = False) -> None:
        sig = self._signature()
        if not force and sig == self._synced:
            return
        tensors = self._tensors()
        s = capi.stream_ptr()
        for name in self._names:
            t = tensors[name]
            if not t.is_cuda:
                raise RuntimeError(f"parameter {name} is on {t.device}; move the module to the GPU first")
            src = t.detach().to(torch.float32).contiguous()
            shape = (C.c_int64 * src.ndim)(*src.shape)
            capi.check(capi.lib.dfot_dit_load_weight(self._handle, name.encode(), capi.ptr(src), shape, src.ndim, s))
        capi.check(capi.lib.dfot_dit_finalize(self._handle, s))
        self._synced = sig
        # captured sampler graphs bake the kernels chosen for the OLD weights (attention variant by score bound) and their pointers
        self.reserve_generation = getattr(self, "reserve_generation", 0) + 1

    def set_option(self, key: str, value: int) -> None:
        self.reserve_generation = getattr(self, "reserve_generation", 0) + 1  # options select kernels: captured graphs are stale
        capi.check(capi.lib.dfot_dit_set_option(self._handle, key.encode(), int(value)))

    def attn_timing(self):
        tot, n = C.c_double(), C.c_int64()
        capi.check(capi.lib.dfot_dit_attn_timing(self._handle, C.byref(tot), C.byref(n)))
        return tot.value, n.value

    def reserve(self, batch: int) -> None:
        if batch > self._reserved:
            torch.cuda.synchronize()
            capi.check(capi.lib.dfot_dit_reserve(self._handle, int(batch)))
            self._reserved = batch
            self.reserve_generation = getattr(self, "reserve_generation", 0) + 1  # workspace pointers changed

    def forward(self, x: torch.Tensor, noise_levels: torch.Tensor, external_cond: Optional[torch.Tensor] = None,
                external_cond_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """BaseBackbone.forward; dispatched as the torch operator ``dfot::dit3d_forward`` (ops.py)."""
        if external_cond is not None:
            raise ValueError("this DiT3D was built without an external condition embedding")
        if self._op_key is None:
            self._op_key = ops.register_model(self)
        params = [p for _, p in self.named_parameters()]
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            # the reference's training_step differentiates through `self.model(...)` (discrete_diffusion.py model_predictions ->
            # accelerator.backward): saved-activation forward + the hand-written backward, registered with autograd (ops.py)
            return torch.ops.dfot.dit3d_forward_train(x, noise_levels, params, self._op_key)
        return torch.ops.dfot.dit3d_forward(x, noise_levels, self._op_key)

    # ------------------------------------------------------------------ training form (autograd)
    def _train_engine(self, params):
        """trainer.DiT3DTrainer on the module's CURRENT weights: built once; its flat parameter buffer is refreshed (and the bf16
        compute copies re-packed) whenever a parameter changed since the last training forward."""
        from . import trainer as _trainer
        sig = tuple((t.data_ptr(), t._version) for t in params)
        if self._trainer is None:
            self._trainer = _trainer.DiT3DTrainer(self.cfg, self.x_shape, self._ctor["max_tokens"], timesteps=self._ctor["timesteps"])
            missing = [n for n in self._trainer.layout if n not in self._train_names]
            if missing:
                raise RuntimeError(f"the training engine expects parameters the module does not have: {missing[:4]}")
        if sig != self._trainer_sig:
            with torch.no_grad():
                self._trainer.load_state_dict({n: t for n, t in zip(self._train_names, params) if n in self._trainer.layout})
            self._trainer_sig = sig
        return self._trainer

    def _train_forward_impl(self, x, noise_levels, params):
        if x.ndim != 5 or tuple(x.shape[2:]) != self.x_shape:
            raise ValueError(f"x has shape {tuple(x.shape)}, expected (B, T, {', '.join(map(str, self.x_shape))})")
        if x.shape[1] > self.max_tokens:
            raise ValueError(f"{x.shape[1]} tokens exceed max_tokens={self.max_tokens}")
        if tuple(noise_levels.shape) != tuple(x.shape[:2]):
            raise ValueError(f"noise_levels has shape {tuple(noise_levels.shape)}, expected {tuple(x.shape[:2])}")
        if noise_levels.is_floating_point():
            raise TypeError("DiT3D takes integer noise levels (DiscreteDiffusion passes the level index)")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError(f"the backbone's parameters are on {dev}; move the module to the GPU first (there is no CPU path)")
        capi.require_device(dev, x=x, noise_levels=noise_levels)
        self._train_stamp += 1
        with torch.no_grad():
            return self._train_engine(params).forward(x, noise_levels).to(x.dtype)

    def _train_backward_impl(self, grad_out, params, stamp=None, want_dx=False):
        eng = self._trainer
        if eng is None:
            raise RuntimeError("backward without a training forward")
        if stamp is not None and stamp != self._train_stamp:
            raise RuntimeError(
                f"DiT3D: backward of training forward #{stamp}, but forward #{self._train_stamp} has run since and overwritten the saved "
                "activations (one engine per module). Run backward after each forward (accumulate gradients as forward/backward pairs).")
        with torch.no_grad():
            eng.backward(grad_out)
            grads = [eng.view(n, eng.grads).to(p.dtype).clone() if n in eng.layout else torch.zeros_like(p)
                     for n, p in zip(self._train_names, params)]
            if want_dx:  # reconstruction guidance differentiates the prediction w.r.t. x_t (discrete_diffusion.py:485-513)
                grads.append(eng.input_grad().to(grad_out.dtype))
            return grads

    def _forward_impl(self, x: torch.Tensor, noise_levels: torch.Tensor) -> torch.Tensor:
        if x.ndim != 5 or tuple(x.shape[2:]) != self.x_shape:
            raise ValueError(f"x has shape {tuple(x.shape)}, expected (B, T, {', '.join(map(str, self.x_shape))})")
        b, t = x.shape[:2]
        if t > self.max_tokens:
            raise ValueError(f"{t} tokens exceed max_tokens={self.max_tokens}")
        if tuple(noise_levels.shape) != (b, t):
            raise ValueError(f"noise_levels has shape {tuple(noise_levels.shape)}, expected {(b, t)}")
        if noise_levels.is_floating_point():
            raise TypeError("DiT3D takes integer noise levels (DiscreteDiffusion passes the level index)")
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError(f"the backbone's parameters are on {dev}; move the module to the GPU first (there is no CPU path)")
        capi.require_device(dev, x=x, noise_levels=noise_levels)
        self.sync_weights()
        self.reserve(b)
        xf = x.detach().to(torch.float32).contiguous()
        kf = noise_levels.detach().to(torch.int32).contiguous()
        out = torch.empty_like(xf)
        capi.check(capi.lib.dfot_dit_forward(self._handle, capi.ptr(xf, torch.float32, "x"), capi.ptr(kf, torch.int32, "noise_levels"),
                                             capi.ptr(out), b, t, capi.stream_ptr()))
        return out.to(x.dtype)

    def read_tap(self, name: str, rows: int) -> torch.Tensor:
        out = torch.empty(rows, self.hidden_size, device="cuda", dtype=torch.float32)
        capi.check(capi.lib.dfot_dit_read_tap(self._handle, name.encode(), capi.ptr(out), out.numel(), capi.stream_ptr()))
        return out

    def __del__(self):
        try:
            if getattr(self, "_handle", None) and self._handle.value:
                capi.lib.dfot_dit_destroy(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass


class DifferenceDiT3D(DiT3D):
    """The bash/k600 backbone: ``difference_dit3d`` with variant ``factorized_matrix_attention`` (per-frame spatial DiT blocks
    alternating with frame-token MatrixDiT blocks), ``pos_emb_type: sinusoidal_2d``, ``merge_type: interleaved``
    (algorithms/dfot/backbones/dit/difference_dit3d.py:12-226; dit_base.py:155-226; dit_blocks.py:211-350,549-652).
    Like the reference class it doubles ``max_tokens``: ``forward`` takes the 2T interleaved (difference, frame) tokens."""

    def _configure(self, c: "capi.DiTConfig", cfg, max_tokens: int) -> None:
        if _get(cfg, "variant") != "factorized_matrix_attention":
            raise ValueError(f"unsupported DifferenceDiT3D variant {_get(cfg, 'variant')!r}: only 'factorized_matrix_attention'")
        if _get(cfg, "pos_emb_type") != "sinusoidal_2d":
            raise ValueError("only pos_emb_type='sinusoidal_2d' is supported")
        if _get(cfg, "merge_type", "interleaved") != "interleaved":
            raise ValueError("only merge_type='interleaved' is supported")
        if _get(cfg, "matrix_block", "matrix") != "matrix" or _get(cfg, "matrix_multi_token", False) or _get(cfg, "fixed_u", None):
            raise ValueError("only matrix_block='matrix' with learned factors and multi_token=False is supported")
        ratio, tratio = _get(cfg, "spatial_mlp_ratio", None), _get(cfg, "mlp_ratio", None)
        if ratio is None:
            raise AssertionError("spatial_mlp_ratio must be specified for matrix attention")
        c.hidden_size = int(_get(cfg, "embed_row_dim"))
        c.max_tokens = 2 * max_tokens  # doubling max_tokens for difference encoding
        c.mlp_hidden = int(c.hidden_size * ratio) if ratio else 0
        c.variant = 1
        c.embed_col_dim = int(_get(cfg, "embed_col_dim"))
        c.num_col_heads = int(_get(cfg, "num_col_heads"))
        c.num_row_heads = int(_get(cfg, "num_row_heads"))
        c.temporal_mlp_hidden = int(c.hidden_size * tratio) if tratio else 0
        c.use_bias = int(bool(_get(cfg, "use_bias")))

    def init_random(self, seed: int = 0) -> None:
        """As DiT3D.init_random; the matrix factors are (in, out) matrices (fan-in = rows), their biases ~ N(0, 0.05^2)."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, t in self._tensors().items():
                leaf = name.rsplit(".", 1)[-1]
                if leaf in ("bias", "qkv_bias", "proj_bias"):
                    v = 0.05 * torch.randn(t.shape, generator=g)
                elif leaf in ("qkv_u", "proj_u", "qkv_v", "proj_v"):
                    v = torch.randn(t.shape, generator=g) / math.sqrt(t.shape[0])
                elif name.startswith("diff_embedder"):
                    v = 0.3 * torch.randn(t.shape, generator=g)
                else:
                    gain = 0.5 if ".modulation." in name else 1.0
                    v = gain * torch.randn(t.shape, generator=g) / math.sqrt(math.prod(t.shape[1:]))
                t.copy_(v.to(t.device))
