"""Training of the DiT3D backbone on the MI355X engine: hand-written forward-with-saved-activations and backward
(``csrc/dit_train.inl``), fused AdamW with gradient-norm clipping on flat fp32 buffers, data parallelism by ONE all-reduce of
the flat gradient buffer per step (RCCL through ``torch.distributed``; the buffer is a torch tensor).

Mirrors, for the DiT3D "full" / rope_3d model (README ``@DiT/XL``, attention-only blocks in this fork):
  * ``DFoTVideo.training_step``                     algorithms/dfot/dfot_video.py:41-75
  * ``DiscreteDiffusion.forward`` (pred_v)          algorithms/dfot/diffusion/discrete_diffusion.py:345-377
  * ``BasePytorchAlgo.configure_optimizers``        AdamW(lr, weight_decay, betas) + Lightning's gradient_clip_val
  * DDP gradient averaging                          experiments (Lightning ``ddp`` strategy)
No autograd and no torch kernels on the path: torch provides device memory, the stream and the collective.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import capi, parallel
from .backbone import _get
from .diffusion import DiffusionConfig, Schedule


class DiT3DTrainer:
    def __init__(self, cfg, x_shape: Sequence[int], max_tokens: int, timesteps: int = 1000,
                 diffusion: Optional[DiffusionConfig] = None, lr: float = 5e-5, weight_decay: float = 0.01,
                 betas: Tuple[float, float] = (0.9, 0.99), eps: float = 1e-8, max_grad_norm: Optional[float] = 1.0,
                 loss_weighting: Optional[Dict] = None):
        self.x_shape = tuple(int(v) for v in x_shape)
        c = capi.DiTConfig()
        c.depth = int(_get(cfg, "depth"))
        c.num_heads = int(_get(cfg, "num_heads"))
        c.patch_size = int(_get(cfg, "patch_size", 2))
        c.in_channels, c.height, c.width = self.x_shape
        c.noise_dim, c.timesteps, c.rope_theta, c.eps = 256, int(timesteps), 10000.0, 1e-6
        ratio = _get(cfg, "spatial_mlp_ratio", None)
        variant = _get(cfg, "variant", "full")
        if variant == "full":  # DiT3D (dit3d.yaml)
            if _get(cfg, "pos_emb_type", "rope_3d") != "rope_3d":
                raise ValueError("DiT3DTrainer builds the 'full' DiT3D with pos_emb_type='rope_3d'")
            c.variant, c.hidden_size, c.max_tokens = 0, int(_get(cfg, "hidden_size")), int(max_tokens)
        elif variant == "factorized_matrix_attention":  # DifferenceDiT3D (bash/k600): as dit_backbone.DifferenceDiT3D._configure
            if _get(cfg, "pos_emb_type") != "sinusoidal_2d" or _get(cfg, "merge_type", "interleaved") != "interleaved":
                raise ValueError("the difference model trains with pos_emb_type='sinusoidal_2d' and merge_type='interleaved'")
            if _get(cfg, "matrix_block", "matrix") != "matrix" or _get(cfg, "matrix_multi_token", False) or _get(cfg, "fixed_u", None):
                raise ValueError("only matrix_block='matrix' with learned factors and multi_token=False is supported")
            if ratio is None:
                raise AssertionError("spatial_mlp_ratio must be specified for matrix attention")
            tratio = _get(cfg, "mlp_ratio", None)
            c.variant, c.hidden_size, c.max_tokens = 1, int(_get(cfg, "embed_row_dim")), 2 * int(max_tokens)
            c.embed_col_dim = int(_get(cfg, "embed_col_dim"))
            c.num_col_heads, c.num_row_heads = int(_get(cfg, "num_col_heads")), int(_get(cfg, "num_row_heads"))
            c.temporal_mlp_hidden = int(c.hidden_size * tratio) if tratio else 0
            c.use_bias = int(bool(_get(cfg, "use_bias")))
        else:
            raise ValueError(f"no training path for DiT variant {variant!r}")
        c.mlp_hidden = int(c.hidden_size * ratio) if ratio else 0
        self._ccfg = c
        self.max_tokens = int(c.max_tokens)
        self._handle = C.c_void_p()
        capi.check(capi.lib.dfot_dit_train_create(C.byref(c), C.byref(self._handle)))
        lib, h = capi.lib, self._handle
        self.numel = int(lib.dfot_dit_train_total_numel(h))
        shape, ndim = (C.c_int64 * 4)(), C.c_int()
        self.layout: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        for i in range(lib.dfot_dit_train_num_params(h)):
            capi.check(lib.dfot_dit_train_param_shape(h, i, shape, C.byref(ndim)))
            self.layout[lib.dfot_dit_train_param_name(h, i).decode()] = (
                int(lib.dfot_dit_train_param_offset(h, i)), tuple(int(shape[k]) for k in range(ndim.value)))
        # flat buffers: torch owns them (the gradient buffer is what torch.distributed all-reduces)
        self.params = torch.zeros(self.numel, device="cuda", dtype=torch.float32)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self._sumsq = torch.zeros(1, device="cuda", dtype=torch.float32)
        capi.check(lib.dfot_dit_train_attach(h, capi.ptr(self.params), capi.ptr(self.grads)))
        self.lr, self.weight_decay, self.betas, self.eps, self.max_grad_norm = lr, weight_decay, tuple(betas), eps, max_grad_norm
        self.step_count = 0
        self.schedule = Schedule(diffusion or DiffusionConfig(beta_schedule="cosine", is_continuous=False, timesteps=timesteps))
        self.loss_weighting = dict(loss_weighting or {})
        self._reserved = 0
        self.ema: Optional[torch.Tensor] = None  # EMAModel shadow weights (flat), updated inside the optimizer kernel
        self.ema_decay = 0.0
        self._acc: Optional[torch.Tensor] = None
        self._acc_n = 0
        self._dirty = True
        self._last: Optional[dict] = None

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            capi.lib.dfot_dit_train_destroy(h)
            self._handle = None

    # ------------------------------------------------------------------ parameters (reference state_dict names)
    def view(self, name: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        off, shape = self.layout[name]
        return (self.params if buf is None else buf)[off: off + int(np.prod(shape))].view(shape)

    def load_state_dict(self, state: Dict[str, torch.Tensor], strict: bool = True) -> None:
        missing = [k for k in self.layout if k not in state]
        extra = [k for k in state if k not in self.layout]
        if strict and (missing or extra):
            raise KeyError(f"state_dict mismatch: missing {missing[:4]}, unexpected {extra[:4]}")
        for k in self.layout:
            if k in state:
                t = state[k]
                if tuple(t.shape) != self.layout[k][1]:
                    raise ValueError(f"{k}: shape {tuple(t.shape)} != {self.layout[k][1]}")
                self.view(k).copy_(t.to(device="cuda", dtype=torch.float32))
        self._dirty = True

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: self.view(k).detach().clone() for k in self.layout}

    def grad_dict(self) -> Dict[str, torch.Tensor]:
        return {k: self.view(k, self.grads).detach().clone() for k in self.layout}

    def _sync(self) -> None:
        if self._dirty:
            capi.check(capi.lib.dfot_dit_train_sync_weights(self._handle, capi.stream_ptr()))
            self._dirty = False

    # ------------------------------------------------------------------ forward / backward (the autograd pair)
    def forward(self, x: torch.Tensor, noise_levels: torch.Tensor) -> torch.Tensor:
        b, t = x.shape[:2]
        if tuple(x.shape[2:]) != self.x_shape:
            raise ValueError(f"x has frame shape {tuple(x.shape[2:])}, expected {self.x_shape}")
        if b > self._reserved:
            capi.check(capi.lib.dfot_dit_train_reserve(self._handle, b))
            self._reserved = b
        self._sync()
        xd = x.to(device="cuda", dtype=torch.float32).contiguous()
        lv = noise_levels.to(device="cuda", dtype=torch.int32).contiguous()
        out = torch.empty_like(xd)
        capi.check(capi.lib.dfot_dit_train_forward(self._handle, capi.ptr(xd), capi.ptr(lv), capi.ptr(out), b, t, capi.stream_ptr()))
        self._keep = (xd, lv)  # the engine reads x again in backward (patch-embedding gradient)
        return out

    def backward(self, d_out: torch.Tensor) -> None:
        g = d_out.to(device="cuda", dtype=torch.float32).contiguous()
        capi.check(capi.lib.dfot_dit_train_backward(self._handle, capi.ptr(g), capi.stream_ptr()))

    def input_grad(self) -> torch.Tensor:
        """d(sum(out * d_out)) / d x of the last forward / backward pair, in x's layout (reconstruction guidance, discrete_diffusion.py:485-513)"""
        dx = torch.empty_like(self._keep[0])
        capi.check(capi.lib.dfot_dit_train_input_grad(self._handle, capi.ptr(dx), capi.stream_ptr()))
        return dx

    # ------------------------------------------------------------------ one training step
    def loss_and_grads(self, xs: torch.Tensor, k: torch.Tensor, noise: torch.Tensor, masks: Optional[torch.Tensor] = None):
        """DiscreteDiffusion.forward (pred_v) + _reweight_loss + backward: noise every token to its level, one forward, the
        weighted v-space error averaged over (B, T) with the loss masks, gradients of every parameter.  Returns the loss (device scalar)."""
        b, t = xs.shape[:2]
        f = int(np.prod(xs.shape[2:]))
        kk = k.detach().cpu().numpy().astype(np.int64)
        sch = self.schedule
        w = sch.loss_weights(kk, **self.loss_weighting).astype(np.float32)
        mk = np.ones((b, t), np.float32) if masks is None else masks.detach().cpu().numpy().astype(np.float32).reshape(b, t)
        tab = np.stack([sch.sqrt_alphas_cumprod[kk], sch.sqrt_one_minus_alphas_cumprod[kk], w, 2.0 * w * mk / (f * b * t)]).astype(np.float32)
        tab = torch.from_numpy(tab).cuda().contiguous()
        x = xs.to(device="cuda", dtype=torch.float32).contiguous()
        eps = noise.to(device="cuda", dtype=torch.float32).clamp(-self.schedule.cfg.clip_noise, self.schedule.cfg.clip_noise).contiguous()
        x_k = torch.empty_like(x)
        s = capi.stream_ptr
        capi.check(capi.lib.dfot_hg_prepare(capi.ptr(x), capi.ptr(eps), capi.ptr(tab[0]), capi.ptr(tab[1]), capi.ptr(x_k), b, 1, t, f, s()))
        v = self.forward(x_k, k)
        per_token = torch.empty(b, t, device="cuda")
        scratch = torch.empty(int(capi.lib.dfot_vpred_loss_scratch_floats(b, t, f)), device="cuda")
        capi.check(capi.lib.dfot_vspace_loss(capi.ptr(x), capi.ptr(eps), capi.ptr(v), capi.ptr(tab[0]), capi.ptr(tab[1]), capi.ptr(tab[2]),
                                             None, capi.ptr(scratch), capi.ptr(per_token), b, t, f, s()))
        dv = torch.empty_like(x)
        capi.check(capi.lib.dfot_vloss_grad(capi.ptr(x), capi.ptr(eps), capi.ptr(v), capi.ptr(tab[0]), capi.ptr(tab[1]), capi.ptr(tab[3]),
                                            capi.ptr(dv), b, t, f, 1, s()))
        self.backward(dv)
        return (per_token * torch.from_numpy(mk).cuda()).mean()

    def difference_loss_and_grads(self, frames: torch.Tensor, k: torch.Tensor, noise: torch.Tensor, masks: Optional[torch.Tensor] = None):
        """DifferenceDFoTVideo.training_step (difference_dfot_video.py:80-105): frame differences (first frame against itself) are
        interleaved with the frames (difference first), noise levels and loss masks are doubled the same way, then the ordinary
        denoising loss on the 2T merged tokens.  frames (B,T,C,H,W), k / masks (B,T), noise (B,2T,C,H,W)."""
        if self._ccfg.variant != 1:
            raise ValueError("difference_loss_and_grads needs the difference model (variant factorized_matrix_attention)")
        fr = frames.to(device="cuda", dtype=torch.float32)
        diff = torch.diff(fr, dim=1, prepend=fr[:, :1])
        merge = lambda a, b: torch.stack([a, b], dim=2).flatten(1, 2)
        kk = k.to("cuda")
        mk = None if masks is None else merge(masks.to("cuda"), masks.to("cuda"))
        return self.loss_and_grads(merge(diff, fr), merge(kk, kk), noise, mk)

    def accumulate(self) -> None:
        """accumulate_grad_batches: add the gradients of the last backward to the running sum used by the next optimizer_step"""
        if self._acc is None:
            self._acc = torch.zeros_like(self.grads)
        self._acc.add_(self.grads)
        self._acc_n += 1

    def optimizer_step(self, world_size: int = 1) -> None:
        """[all-reduce + average the flat gradient buffer] -> global-norm clip -> AdamW -> refresh the bf16 compute weights"""
        if self._acc_n:
            self.grads.copy_(self._acc).mul_(1.0 / self._acc_n)
            self._acc.zero_()
            self._acc_n = 0
        if world_size > 1:
            parallel.allreduce_mean_(self.grads)
        self.step_count += 1
        s = capi.stream_ptr
        sumsq = None
        if self.max_grad_norm is not None:
            capi.check(capi.lib.dfot_sumsq(capi.ptr(self.grads), self.numel, capi.ptr(self._sumsq), s()))
            sumsq = self._sumsq
        capi.check(capi.lib.dfot_adamw_step(capi.ptr(self.params), capi.ptr(self.grads), capi.ptr(self.exp_avg), capi.ptr(self.exp_avg_sq),
                                            self.numel, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count,
                                            capi.ptr(sumsq), float(self.max_grad_norm or 0.0), capi.ptr(self.ema), float(self.ema_decay), s()))
        self._dirty = True

    def training_step(self, xs: torch.Tensor, k: torch.Tensor, noise: torch.Tensor, masks: Optional[torch.Tensor] = None,
                      world_size: int = 1) -> torch.Tensor:
        loss = self.loss_and_grads(xs, k, noise, masks)
        self.optimizer_step(world_size)
        return loss

    # ------------------------------------------------------------------ EMA and optimizer state (checkpoint / resume)
    def enable_ema(self, decay: float) -> None:
        """experiment.ema (algorithms/common/ema.py): shadow weights start as a copy of the parameters"""
        self.ema, self.ema_decay = self.params.clone(), float(decay)

    def ema_state_dict(self) -> Dict[str, torch.Tensor]:
        if self.ema is None:
            raise RuntimeError("EMA is not enabled")
        return {k: self.view(k, self.ema).detach().clone() for k in self.layout}

    def optimizer_state_dict(self) -> Dict:
        """torch.optim.AdamW.state_dict() layout (parameter index = position in the reference's parameter order)"""
        state = {i: {"step": torch.tensor(float(self.step_count)), "exp_avg": self.view(k, self.exp_avg).clone(),
                     "exp_avg_sq": self.view(k, self.exp_avg_sq).clone()} for i, k in enumerate(self.layout)} if self.step_count else {}
        group = dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, amsgrad=False,
                     params=list(range(len(self.layout))))
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: Dict) -> None:
        names = list(self.layout)
        steps = set()
        for i, st in sd.get("state", {}).items():
            k = names[int(i)]
            self.view(k, self.exp_avg).copy_(st["exp_avg"].to("cuda"))
            self.view(k, self.exp_avg_sq).copy_(st["exp_avg_sq"].to("cuda"))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ: the flat optimizer keeps one")
        self.step_count = steps.pop() if steps else 0
        if sd.get("param_groups"):
            g0 = sd["param_groups"][0]
            self.lr, self.betas, self.eps, self.weight_decay = g0["lr"], tuple(g0["betas"]), g0["eps"], g0["weight_decay"]

    def grad_norm(self) -> float:
        capi.check(capi.lib.dfot_sumsq(capi.ptr(self.grads), self.numel, capi.ptr(self._sumsq), capi.stream_ptr()))
        return float(self._sumsq.sqrt().item())
