"""Drop-in UViT3DPose backbone backed by libdfot_hip.so.

Mirrors the reference's plugin contract for this path:
  * constructor keywords of DiscreteDiffusion._build_model
    (algorithms/dfot/diffusion/discrete_diffusion.py:84-92)
  * ``forward(x, noise_levels, external_cond, external_cond_mask)`` of BaseBackbone
    (algorithms/dfot/backbones/base_backbone.py:78-86, u_vit/u_vit3d_pose.py:63-131)
  * state-dict key names / shapes, so reference checkpoints load with
    ``load_state_dict``.
Parameters live here as fp32 ``nn.Parameter``s (the reference layout); the C library keeps
packed bf16 copies that are refreshed whenever a parameter changes.
Under ``torch.no_grad()`` (sampling) ``forward`` runs the fused inference engine (per-window pose / FiLM caches).  With gradients
enabled and trainable parameters it runs the training form -- saved-activation forward + hand-written backward
(uvit_train.UViT3DPoseTrainer) behind ``dfot::uvit3d_pose_forward_train`` -- so ``loss.backward()`` / ``accelerator.backward(loss)``
fill ``param.grad`` exactly as the reference's autograd does (continuous_diffusion.py:154, simple_video_generation.py:260-270).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import nn

from . import capi, ops


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default) if not hasattr(cfg, "get") else cfg.get(key, default)


class _Node(nn.Module):
    """Anonymous container so that dotted reference key names map onto a module tree."""


class UViT3DPose(nn.Module):
    def __init__(self, cfg, x_shape: Sequence[int], max_tokens: int, external_cond_dim: int = 0,
                 external_cond_type: str = "action", external_cond_num_classes: Optional[int] = None,
                 use_causal_mask: bool = False, **kwargs):
        super().__init__()
        block_types = list(_get(cfg, "block_types", ["ResBlock", "ResBlock", "TransformerBlock", "TransformerBlock"]))
        if block_types != ["ResBlock", "ResBlock", "TransformerBlock", "TransformerBlock"]:
            raise ValueError(f"unsupported block_types {block_types}")
        if _get(cfg, "pos_emb_type", "rope") != "rope":
            raise ValueError("only pos_emb_type='rope' is supported")
        if not _get(cfg, "use_fourier_noise_embedding", True):
            raise ValueError("only the Fourier noise-level embedding (continuous diffusion) is supported")
        cond = _get(cfg, "conditioning", None)
        cond_dim = _get(cond, "dim", None) if cond is not None else None
        self.cfg = cfg
        self.x_shape = tuple(x_shape)
        self.max_tokens = self.temporal_length = int(max_tokens)
        self.external_cond_dim = int(cond_dim or 180)
        self.use_causal_mask = use_causal_mask
        c = capi.UViTConfig()
        c.channels[:] = list(_get(cfg, "channels"))
        c.emb_channels = int(_get(cfg, "emb_channels"))
        c.num_updown_blocks[:] = list(_get(cfg, "num_updown_blocks"))
        c.num_mid_blocks = int(_get(cfg, "num_mid_blocks"))
        c.num_heads = int(_get(cfg, "num_heads"))
        c.in_channels = int(self.x_shape[0])
        c.resolution = int(self.x_shape[-1])
        c.max_tokens = self.max_tokens
        c.cond_dim = self.external_cond_dim
        c.noise_dim = 256
        c.rope_theta = 10000.0
        c.eps = 1e-6
        if int(_get(cfg, "patch_size", 2)) != 2:
            raise ValueError("only patch_size=2 is supported")
        self._ccfg = c
        self._handle = C.c_void_p()
        capi.check(capi.lib.dfot_uvit_create(C.byref(c), C.byref(self._handle)))
        self._names = []
        self._persistent_buffers = {"noise_level_pos_embedding.timesteps.freqs",
                                    "noise_level_pos_embedding.timesteps.phases"}
        shape = (C.c_int64 * 4)()
        ndim = C.c_int()
        for i in range(capi.lib.dfot_uvit_num_params(self._handle)):
            name = capi.lib.dfot_uvit_param_name(self._handle, i).decode()
            capi.check(capi.lib.dfot_uvit_param_shape(self._handle, i, shape, C.byref(ndim)))
            self._register(name, tuple(shape[k] for k in range(ndim.value)))
            self._names.append(name)
        self._synced: Optional[Tuple] = None
        self._reserved = 0
        self._op_key: Optional[int] = None
        self._cond_key = None
        self._cond_refs = None
        self._train_names = None     # parameter names in named_parameters() order (operator input order of the training form)
        self._trainer = None         # uvit_train.UViT3DPoseTrainer on this module's weights, built at the first training forward
        self._trainer_sig = None
        self.live_frames = None      # see _forward_impl: set by the sampler around its backbone calls, None otherwise
        self.fresh_frames = None
        self._fresh_key = None       # conditioning key of the previous forward (frozen frames are only honoured behind the same one)
        self._train_stamp = 0        # counts training forwards: the autograd ctx of a forward remembers its number (ops._train_setup_context)
        self._dropout_generator: Optional[torch.Generator] = None  # set a CUDA generator to enable the MLP-branch nn.Dropout in train()

    # ------------------------------------------------------------------ module tree
    def _register(self, name: str, shape: Tuple[int, ...]) -> None:
        *path, leaf = name.split(".")
        node: nn.Module = self
        for part in path:
            if part not in node._modules:
                node.add_module(part, _Node())
            node = node._modules[part]
        t = torch.zeros(shape, dtype=torch.float32)
        if name in self._persistent_buffers:
            node.register_buffer(leaf, t, persistent=True)
        else:
            node.register_parameter(leaf, nn.Parameter(t))

    def _tensors(self) -> Dict[str, torch.Tensor]:
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        return sd

    def reset_parameters(self, seed: int = 0) -> None:
        """Reference-style default init (zero-initialised output projections included)."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, t in self._tensors().items():
                leaf = name.rsplit(".", 1)[-1]
                if leaf == "freqs":
                    t.copy_(2 * math.pi * torch.randn(t.shape, generator=g))
                elif leaf == "phases":
                    t.copy_(2 * math.pi * torch.rand(t.shape, generator=g))
                elif any(s in name for s in (".attn_out.", ".mlp_out.2.", ".out_rest.1.", "project_output")):
                    t.zero_()
                elif leaf == "bias":
                    t.zero_()
                elif t.ndim == 1:
                    t.fill_(1.0)
                else:
                    fan_in = math.prod(t.shape[1:])
                    bound = 1.0 / math.sqrt(fan_in)
                    t.copy_((torch.rand(t.shape, generator=g) * 2 - 1) * bound)

    # ------------------------------------------------------------------ weights -> engine
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
            capi.check(capi.lib.dfot_uvit_load_weight(self._handle, name.encode(), capi.ptr(src), shape, src.ndim, s))
        capi.check(capi.lib.dfot_uvit_finalize(self._handle, s))
        self._synced = sig
        # captured sampler graphs bake the kernels chosen for the OLD weights (attention variant by score bound) and their pointers
        self.reserve_generation = getattr(self, "reserve_generation", 0) + 1

    def set_option(self, key: str, value: int) -> None:
        self.reserve_generation = getattr(self, "reserve_generation", 0) + 1  # options select kernels: captured graphs are stale
        capi.check(capi.lib.dfot_uvit_set_option(self._handle, key.encode(), int(value)))

    def query(self, key: str) -> float:
        """read-outs of the engine ("score_bound_l2", "attn_kernel_l2": include/dfot_hip.h); the weights are synced first"""
        self.sync_weights()
        val = C.c_double()
        capi.check(capi.lib.dfot_uvit_query(self._handle, key.encode(), C.byref(val)))
        return float(val.value)

    def attn_timing(self):
        """(total_ms, launches) of the level-2 attention launches recorded since set_option('time_attn', N)."""
        tot, n = C.c_double(), C.c_int64()
        capi.check(capi.lib.dfot_uvit_attn_timing(self._handle, C.byref(tot), C.byref(n)))
        return tot.value, n.value

    def init_random(self, seed: int = 0, zero_init_scale: float = 0.3) -> None:
        """Non-degenerate random weights for benchmarks (the reference's default init zeroes every output
        projection, which would make all activations trivial): weights ~ N(0, 1/fan_in), biases ~ N(0, 0.02^2),
        gains ~ 1 + N(0, 0.1^2), Fourier freqs 2*pi*N(0,1), phases 2*pi*U(0,1)."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, t in self._tensors().items():
                leaf = name.rsplit(".", 1)[-1]
                if leaf == "freqs":
                    v = 2 * math.pi * torch.randn(t.shape, generator=g)
                elif leaf == "phases":
                    v = 2 * math.pi * torch.rand(t.shape, generator=g)
                elif leaf == "bias":
                    v = 0.02 * torch.randn(t.shape, generator=g)
                elif t.ndim == 1:
                    v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
                else:
                    fan_in = t.shape[0] if name.startswith("project_output") else math.prod(t.shape[1:])
                    v = torch.randn(t.shape, generator=g) / math.sqrt(fan_in)
                    if any(s in name for s in (".attn_out.", ".mlp_out.2.", ".out_rest.1.", "project_output")):
                        v = v * zero_init_scale
                t.copy_(v.to(t.device))

    def reserve(self, batch: int) -> None:
        if batch > self._reserved:
            torch.cuda.synchronize()
            capi.check(capi.lib.dfot_uvit_reserve(self._handle, int(batch)))
            self._reserved = batch
            self.reserve_generation = getattr(self, "reserve_generation", 0) + 1  # workspace pointers changed
            self._cond_key = None

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, noise_levels: torch.Tensor, external_cond: Optional[torch.Tensor] = None,
                external_cond_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """BaseBackbone.forward; dispatched as the torch operator ``dfot::uvit3d_pose_forward`` (ops.py)."""
        assert external_cond is not None, "External condition (camera pose) is required for U-ViT3DPose model."
        if self._op_key is None:
            self._op_key = ops.register_model(self)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if self._train_names is None:
                self._train_names = [n for n, _ in self.named_parameters()]
            params = [p for _, p in self.named_parameters()]
            return torch.ops.dfot.uvit3d_pose_forward_train(x, noise_levels, external_cond, external_cond_mask, params, self._op_key)
        return torch.ops.dfot.uvit3d_pose_forward(x, noise_levels, external_cond, external_cond_mask, self._op_key)

    # ------------------------------------------------------------------ training form (autograd)
    def _train_engine(self, params):
        """The saved-activation engine on the module's CURRENT weights: built once, its flat parameter buffer refreshed (and the
        bf16 operand copies re-packed) whenever a parameter changed since the last training forward."""
        from . import uvit_train
        sig = tuple((t.data_ptr(), t._version) for t in params) + tuple((b.data_ptr(), b._version) for _, b in self.named_buffers())
        if self._trainer is None:
            cfg = dict(channels=list(self._ccfg.channels), emb_channels=int(self._ccfg.emb_channels), patch_size=2,
                       block_types=["ResBlock", "ResBlock", "TransformerBlock", "TransformerBlock"],
                       num_updown_blocks=list(self._ccfg.num_updown_blocks), num_mid_blocks=int(self._ccfg.num_mid_blocks),
                       num_heads=int(self._ccfg.num_heads), in_channels=int(self._ccfg.in_channels), resolution=int(self._ccfg.resolution),
                       max_tokens=self.max_tokens, cond_dim=self.external_cond_dim, noise_dim=int(self._ccfg.noise_dim),
                       eps=float(self._ccfg.eps), rope_theta=float(self._ccfg.rope_theta),
                       block_dropouts=list(_get(self.cfg, "block_dropouts", [0.0] * 4) or [0.0] * 4))
            # the reference's parameter order (names as registered) + the persistent Fourier buffers
            sd = {n: t for n, t in zip(self._train_names, params)}
            sd.update(dict(self.named_buffers()))
            ordered = {n: sd[n] for n in self._names}
            self._trainer = uvit_train.UViT3DPoseTrainer(ordered, cfg)
            self._trainer_sig = sig
        elif sig != self._trainer_sig:
            with torch.no_grad():
                for n, t in zip(self._train_names, params):
                    self._trainer.p[n].copy_(t)
                for n, b in self.named_buffers():
                    self._trainer.p[n].copy_(b)
            self._trainer.sync()
            self._trainer_sig = sig
        return self._trainer

    def _train_forward_impl(self, x, noise_levels, external_cond, external_cond_mask, params):
        assert x.shape[1] == self.temporal_length, (
            f"Temporal length of U-ViT is set to {self.temporal_length}, but input has temporal length {x.shape[1]}.")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError(f"the backbone's parameters are on {dev}; move the module to the GPU first (there is no CPU path)")
        capi.require_device(dev, x=x, noise_levels=noise_levels, external_cond=external_cond, external_cond_mask=external_cond_mask)
        eng = self._train_engine(params)
        # training-time pose dropout (RandomDropoutPatchEmbed, embeddings.py:390-428): per-video Bernoulli(external_cond_dropout) while
        # the module is in train() mode; an explicit mask (inference-style call under grad) is honoured as given
        drop = external_cond_mask
        pdrop = float(_get(self.cfg, "external_cond_dropout", 0.0) or 0.0)
        if drop is None and self.training and pdrop > 0:
            drop = torch.rand(x.shape[0], device=dev) < pdrop
        eng.dropout_generator = self._dropout_generator if self.training else None
        self._train_stamp += 1  # this forward now owns the engine's saved activations (checked by the backward)
        with torch.no_grad():
            return eng.forward(x, noise_levels, external_cond, drop).to(x.dtype)

    def _train_backward_impl(self, grad_out, params, stamp=None, want_dx=False):
        eng = self._trainer
        if eng is None:
            raise RuntimeError("backward without a training forward")
        if stamp is not None and stamp != self._train_stamp:
            raise RuntimeError(
                f"UViT3DPose: backward of training forward #{stamp}, but forward #{self._train_stamp} has run since and overwritten the "
                "saved activations (one engine per module). Run backward after each forward (accumulate gradients as forward/backward "
                "pairs); two forwards of one module inside one loss are not supported.")
        with torch.no_grad():
            grads = eng.backward(grad_out, input_grad=want_dx)
        out = [grads[n].to(p.dtype).reshape(p.shape) if n in grads else torch.zeros_like(p) for n, p in zip(self._train_names, params)]
        if want_dx:  # the gradient w.r.t. x rides as the last element (ops.py hands it to autograd as x's gradient)
            out.append(eng.dx_in.to(grad_out.dtype))
        return out

    def _forward_impl(self, x: torch.Tensor, noise_levels: torch.Tensor, external_cond: Optional[torch.Tensor] = None,
                      external_cond_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert x.shape[1] == self.temporal_length, (
            f"Temporal length of U-ViT is set to {self.temporal_length}, but input has temporal length {x.shape[1]}.")
        assert external_cond is not None, "External condition (camera pose) is required for U-ViT3DPose model."
        b = x.shape[0]
        if tuple(x.shape[2:]) != self.x_shape:
            raise ValueError(f"x has frame shape {tuple(x.shape[2:])}, expected {self.x_shape}")
        if tuple(external_cond.shape) != (b, self.temporal_length, self.external_cond_dim, *self.x_shape[1:]):
            raise ValueError(f"external_cond has shape {tuple(external_cond.shape)}")
        if external_cond_mask is not None:
            assert external_cond_mask.ndim == 1, "embedding mask should be of shape (B,)"
        if tuple(noise_levels.shape) != (b, self.temporal_length):
            raise ValueError(f"noise_levels has shape {tuple(noise_levels.shape)}, expected {(b, self.temporal_length)}")
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError(f"the backbone's parameters are on {dev}; move the module to the GPU first (there is no CPU path)")
        # the kernels dereference raw pointers: a host tensor must be refused before anything is launched
        capi.require_device(dev, x=x, noise_levels=noise_levels, external_cond=external_cond, external_cond_mask=external_cond_mask)
        self.sync_weights()
        self.reserve(b)
        xf = x.detach().to(torch.float32).contiguous()
        kf = noise_levels.detach().to(torch.float32).contiguous()
        # the pose caches are rebuilt only when the conditioning tensors change (the sampler passes the same
        # tensors for all DDIM steps of a window)
        key = (external_cond.data_ptr(), external_cond._version, tuple(external_cond.shape), self._synced,
               None if external_cond_mask is None else (external_cond_mask.data_ptr(), external_cond_mask._version))
        if key != self._cond_key:
            cf = external_cond.detach().to(torch.float32).contiguous()
            mf = None if external_cond_mask is None else external_cond_mask.to(torch.uint8).contiguous()
            capi.check(capi.lib.dfot_uvit_set_conditions(self._handle, capi.ptr(cf, torch.float32, "external_cond"),
                                                         capi.ptr(mf, torch.uint8, "external_cond_mask"), b, capi.stream_ptr()))
            self._cond_key = key
            self._cond_refs = (external_cond, external_cond_mask)  # keep the keyed tensors alive
        out = torch.empty_like(xf)
        live = self.live_frames
        if live is not None:
            # sampler-only hint (sampler.py): uint8 (B, T), 0 = the caller discards this frame's output -> it is not computed past the
            # last transformer block and its rows of the result are zeros.  A plain forward(x, k, cond, mask) never sets it.
            if tuple(live.shape) != (b, self.temporal_length) or live.dtype != torch.uint8 or not live.is_contiguous():
                raise ValueError(f"live_frames must be a contiguous uint8 tensor of shape {(b, self.temporal_length)}")
            capi.require_device(dev, live_frames=live)
        fresh = self.fresh_frames
        if fresh is not None:
            # sampler-only hint: uint8 (B, T), 0 = this frame's input / level / conditioning equal the previous forward's -> its down-path
            # activations are still in the workspace (include/dfot_hip.h, dfot_uvit_forward_cached_masks)
            if tuple(fresh.shape) != (b, self.temporal_length) or fresh.dtype != torch.uint8 or not fresh.is_contiguous():
                raise ValueError(f"fresh_frames must be a contiguous uint8 tensor of shape {(b, self.temporal_length)}")
            capi.require_device(dev, fresh_frames=fresh)
            if key != self._fresh_key:  # the conditioning caches were rebuilt or another batch ran in between: nothing is frozen
                fresh = None
        capi.check(capi.lib.dfot_uvit_forward_cached_masks(self._handle, capi.ptr(xf, torch.float32, "x"),
                                                           capi.ptr(kf, torch.float32, "noise_levels"), capi.ptr(out), b,
                                                           capi.ptr(live, torch.uint8, "live_frames"),
                                                           capi.ptr(fresh, torch.uint8, "fresh_frames"), capi.stream_ptr()))
        self._fresh_key = key
        return out.to(x.dtype)

    def read_tap(self, name: str, channels: int, level: int, batch: int) -> torch.Tensor:
        r = self.x_shape[-1] // 2 // (2 ** level)
        out = torch.empty(batch * self.temporal_length, channels, r, r, device="cuda", dtype=torch.float32)
        capi.check(capi.lib.dfot_uvit_read_tap(self._handle, name.encode(), capi.ptr(out), out.numel(), capi.stream_ptr()))
        return out

    def __del__(self):
        try:
            if getattr(self, "_handle", None) and self._handle.value:
                capi.lib.dfot_uvit_destroy(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass
