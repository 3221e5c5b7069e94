"""UViT3DPose backbone forward (oracle, CPU fp32, functional over a state dict).

Restates (state-dict key names are the reference's):
  * UViT3DPose.forward                       -- algorithms/dfot/backbones/u_vit/u_vit3d_pose.py:63-131
  * UViT3D wiring (levels, up/down, rope)    -- u_vit/u_vit3d.py:30-185
  * ResBlock / TransformerBlock / NormalizeWithCond / Down / Upsample / Embed / Project
                                             -- u_vit/u_vit_blocks.py:16-314
  * RMSNorm                                  -- backbones/modules/normalization.py:5-53
  * Fourier noise-level embedding + MLP      -- backbones/modules/embeddings.py:67-110
    (diffusers==0.32.2 TimestepEmbedding: linear_1 -> SiLU -> linear_2)
  * RotaryEmbedding3D                        -- embeddings.py:156-277
    (rotary_embedding_torch==0.8.6 rotate_half: interleaved pairs (x1,x2)->(-x2,x1))
  * RandomDropoutPatchEmbed (inference mask) -- embeddings.py:336-428
    (timm==1.0.17 PatchEmbed: Conv2d k=s=patch, flatten=False)
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]
USE_SDPA = False  # True: F.scaled_dot_product_attention instead of the explicit softmax (same maths, fused kernel on a GPU)
# True: every ResBlock / TransformerBlock runs under torch.utils.checkpoint (what the reference does per level with use_checkpointing,
# u_vit3d.py:237-243): identical values and gradients, activations of one block at a time -- lets the full-size training-parity test
# (256 x 256, full depth, fp32) run autograd through this restatement within the GPU's memory
CHECKPOINT_BLOCKS = False


@dataclass
class UViTConfig:
    channels: Sequence[int] = (128, 256, 576, 1152)
    emb_channels: int = 1024
    patch_size: int = 2
    block_types: Sequence[str] = ("ResBlock", "ResBlock", "TransformerBlock", "TransformerBlock")
    num_updown_blocks: Sequence[int] = (3, 3, 6)
    num_mid_blocks: int = 20
    num_heads: int = 9
    in_channels: int = 3
    resolution: int = 256
    max_tokens: int = 8
    cond_dim: int = 180
    noise_dim: int = 256
    rope_theta: float = 10000.0
    gn_groups: int = 32
    eps: float = 1e-6

    @property
    def num_levels(self) -> int:
        return len(self.channels)

    def level_res(self, lvl: int) -> int:
        return self.resolution // self.patch_size // (2 ** lvl)


# --------------------------------------------------------------------------- #
# parameter inventory (names + shapes) -- also used by tests to build seeded weights
# --------------------------------------------------------------------------- #
def _res_block_shapes(prefix: str, c: int, e: int) -> Dict[str, tuple]:
    return {
        f"{prefix}.emb_layer.weight": (2 * c, e, 1, 1), f"{prefix}.emb_layer.bias": (2 * c,),
        f"{prefix}.in_layers.0.weight": (c,), f"{prefix}.in_layers.0.bias": (c,),
        f"{prefix}.in_layers.2.weight": (c, c, 3, 3), f"{prefix}.in_layers.2.bias": (c,),
        f"{prefix}.out_norm.weight": (c,), f"{prefix}.out_norm.bias": (c,),
        f"{prefix}.out_rest.1.weight": (c, c, 3, 3), f"{prefix}.out_rest.1.bias": (c,),
    }


def _tr_block_shapes(prefix: str, c: int, e: int, heads: int) -> Dict[str, tuple]:
    d = c // heads
    return {
        f"{prefix}.norm.emb_layer.weight": (2 * c, e), f"{prefix}.norm.emb_layer.bias": (2 * c,),
        f"{prefix}.norm.norm.weight": (c,),
        f"{prefix}.fused_attn_mlp_proj.weight": (7 * c, c), f"{prefix}.fused_attn_mlp_proj.bias": (7 * c,),
        f"{prefix}.q_norm.weight": (d,), f"{prefix}.k_norm.weight": (d,),
        f"{prefix}.attn_out.weight": (c, c), f"{prefix}.attn_out.bias": (c,),
        f"{prefix}.mlp_out.2.weight": (c, 4 * c), f"{prefix}.mlp_out.2.bias": (c,),
    }


def param_shapes(cfg: UViTConfig) -> Dict[str, tuple]:
    """Every persistent state-dict entry of the reference module, in a fixed order."""
    e, ch, p = cfg.emb_channels, list(cfg.channels), cfg.patch_size
    out: Dict[str, tuple] = {
        "noise_level_pos_embedding.timesteps.freqs": (cfg.noise_dim,),
        "noise_level_pos_embedding.timesteps.phases": (cfg.noise_dim,),
        "noise_level_pos_embedding.embedding.linear_1.weight": (e, cfg.noise_dim),
        "noise_level_pos_embedding.embedding.linear_1.bias": (e,),
        "noise_level_pos_embedding.embedding.linear_2.weight": (e, e),
        "noise_level_pos_embedding.embedding.linear_2.bias": (e,),
        "external_cond_embedding.patch_embedder.proj.weight": (e, cfg.cond_dim, p, p),
        "external_cond_embedding.patch_embedder.proj.bias": (e,),
        "embed_input.proj.weight": (ch[0], cfg.in_channels, p, p),
        "embed_input.proj.bias": (ch[0],),
        "project_output.proj.weight": (ch[0], cfg.in_channels, p, p),
        "project_output.proj.bias": (cfg.in_channels,),
    }

    def block(prefix, lvl):
        if cfg.block_types[lvl] == "ResBlock":
            return _res_block_shapes(prefix, ch[lvl], e)
        return _tr_block_shapes(prefix, ch[lvl], e, cfg.num_heads)

    for lvl, n in enumerate(cfg.num_updown_blocks):
        for i in range(n):
            out.update(block(f"down_blocks.{lvl}.{i}", lvl))
        out[f"down_blocks.{lvl}.{n}.conv.weight"] = (ch[lvl + 1], ch[lvl], 3, 3)
        out[f"down_blocks.{lvl}.{n}.conv.bias"] = (ch[lvl + 1],)
    for i in range(cfg.num_mid_blocks):
        out.update(block(f"mid_blocks.{i}", cfg.num_levels - 1))
    for j, lvl in enumerate(reversed(range(cfg.num_levels - 1))):
        out[f"up_blocks.{j}.0.conv.weight"] = (ch[lvl], ch[lvl + 1], 3, 3)
        out[f"up_blocks.{j}.0.conv.bias"] = (ch[lvl],)
        for i in range(cfg.num_updown_blocks[lvl]):
            out.update(block(f"up_blocks.{j}.{i + 1}", lvl))
    return out


def seeded_params(cfg: UViTConfig, seed: int = 0, zero_init_scale: float = 0.3) -> Params:
    """Deterministic non-degenerate weights (the reference zero-initialises its output
    projections, which would make every golden identically zero -- SURVEY.md 8c-4).
    One generator, parameters filled in param_shapes() order:
      weights ~ N(0, 1/fan_in) (x zero_init_scale for the reference's zero-init layers),
      biases ~ N(0, 0.02^2), norm gains ~ 1 + N(0, 0.1^2), Fourier freqs 2*pi*N(0,1),
      phases 2*pi*U(0,1).
    tools/make_golden.py loads exactly these tensors into the reference module."""
    g = torch.Generator().manual_seed(seed)
    out: Params = {}
    for name, shape in param_shapes(cfg).items():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "freqs":
            t = 2 * math.pi * torch.randn(shape, generator=g)
        elif leaf == "phases":
            t = 2 * math.pi * torch.rand(shape, generator=g)
        elif leaf == "bias":
            std = 0.1 if (".in_layers.0." in name or ".out_norm." in name) else 0.02
            t = std * torch.randn(shape, generator=g)
        elif len(shape) == 1:  # norm gains
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            if name.startswith("project_output"):
                fan_in = shape[0]
            else:
                fan_in = math.prod(shape[1:])
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
            if any(s in name for s in (".attn_out.", ".mlp_out.2.", ".out_rest.1.", "project_output")):
                t = t * zero_init_scale
        out[name] = t.to(torch.float32)
    return out


# --------------------------------------------------------------------------- #
# building blocks
# --------------------------------------------------------------------------- #
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).type_as(x) * weight


def rope3d_angles(head_dim: int, sizes: Sequence[int], theta: float) -> torch.Tensor:
    """(T*H*W, head_dim) rotation angles; per-axis share of the head dim follows
    RotaryEmbedding3D's split rule (embeddings.py:264-277); angle duplicated per pair."""
    half = head_dim // 2
    q, r = divmod(half, 3)
    parts = {0: (q, q, q), 1: (q + 1, q, q), 2: (q, q + 1, q + 1)}[r]
    dims = [2 * p for p in parts]
    cols: List[torch.Tensor] = []
    for axis, (dim, n) in enumerate(zip(dims, sizes)):
        inv = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        ang = torch.arange(n, dtype=inv.dtype)[:, None] * inv[None, :]
        ang = ang.repeat_interleave(2, dim=-1)  # (n, dim)
        view = [1, 1, 1, dim]
        view[axis] = n
        cols.append(ang.view(*view).expand(*sizes, dim))
    return torch.cat(cols, dim=-1).reshape(-1, head_dim)


def apply_rope(x: torch.Tensor, ang: torch.Tensor) -> torch.Tensor:
    pairs = x.reshape(*x.shape[:-1], -1, 2)
    rot = torch.stack([-pairs[..., 1], pairs[..., 0]], dim=-1).reshape(x.shape)
    return x * ang.cos() + rot * ang.sin()


def noise_level_embedding(p: Params, k: torch.Tensor) -> torch.Tensor:
    pre = "noise_level_pos_embedding."
    y = k.to(torch.float32)[..., None] * p[pre + "timesteps.freqs"].float()
    y = (y + p[pre + "timesteps.phases"].float()).cos() * math.sqrt(2.0)
    y = F.linear(y, p[pre + "embedding.linear_1.weight"], p[pre + "embedding.linear_1.bias"])
    return F.linear(F.silu(y), p[pre + "embedding.linear_2.weight"], p[pre + "embedding.linear_2.bias"])


def res_block(p: Params, pre: str, x: torch.Tensor, emb: torch.Tensor, cfg: UViTConfig) -> torch.Tensor:
    h = F.group_norm(x, cfg.gn_groups, p[pre + ".in_layers.0.weight"], p[pre + ".in_layers.0.bias"], cfg.eps)
    h = F.conv2d(F.silu(h), p[pre + ".in_layers.2.weight"], p[pre + ".in_layers.2.bias"], padding=1)
    film = F.conv2d(emb, p[pre + ".emb_layer.weight"], p[pre + ".emb_layer.bias"])
    scale, shift = film.chunk(2, dim=1)
    h = F.group_norm(h, cfg.gn_groups, p[pre + ".out_norm.weight"], p[pre + ".out_norm.bias"], cfg.eps)
    h = F.silu(h * (1 + scale) + shift)
    h = F.conv2d(h, p[pre + ".out_rest.1.weight"], p[pre + ".out_rest.1.bias"], padding=1)
    return x + h


def transformer_block(p: Params, pre: str, x: torch.Tensor, emb: torch.Tensor, ang: torch.Tensor,
                      cfg: UViTConfig, taps: Optional[dict] = None, mlp_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mlp_mask (training only): the realised nn.Dropout mask of mlp_out (u_vit_blocks.py:230-234), 0 or 1 / (1 - p) per element"""
    b, n, c = x.shape
    hds = cfg.num_heads
    d = c // hds
    film = F.linear(emb, p[pre + ".norm.emb_layer.weight"], p[pre + ".norm.emb_layer.bias"])
    scale, shift = film.chunk(2, dim=-1)
    xn = rms_norm(x, p[pre + ".norm.norm.weight"], cfg.eps) * (1 + scale) + shift
    fused = F.linear(xn, p[pre + ".fused_attn_mlp_proj.weight"], p[pre + ".fused_attn_mlp_proj.bias"])
    qkv, mlp_h = fused.split([3 * c, 4 * c], dim=-1)
    q, k, v = qkv.view(b, n, 3, hds, d).permute(2, 0, 3, 1, 4)
    q = apply_rope(rms_norm(q, p[pre + ".q_norm.weight"], cfg.eps), ang)
    k = apply_rope(rms_norm(k, p[pre + ".k_norm.weight"], cfg.eps), ang)
    if USE_SDPA:  # the reference's own call (u_vit_blocks.py:264-268); used when the oracle is TIMED on a GPU
        att = F.scaled_dot_product_attention(q, k.to(q.dtype), v.to(q.dtype))
    else:
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(d), dim=-1) @ v
    att = att.permute(0, 2, 1, 3).reshape(b, n, c)
    if taps is not None:
        taps.update(xn=xn, q=q, k=k, v=v, att=att, mlp_h=mlp_h)
    y = x + F.linear(att, p[pre + ".attn_out.weight"], p[pre + ".attn_out.bias"])
    act = F.silu(mlp_h) if mlp_mask is None else F.silu(mlp_h) * mlp_mask
    return y + F.linear(act, p[pre + ".mlp_out.2.weight"], p[pre + ".mlp_out.2.bias"])


# --------------------------------------------------------------------------- #
# whole backbone
# --------------------------------------------------------------------------- #
def forward(p: Params, cfg: UViTConfig, x: torch.Tensor, noise_levels: torch.Tensor,
            external_cond: torch.Tensor, external_cond_mask: Optional[torch.Tensor] = None,
            taps: Optional[dict] = None) -> torch.Tensor:
    """x (B,T,C,H,W), noise_levels (B,T) float, external_cond (B,T,180,H,W),
    external_cond_mask None | bool (B,) (True => that video's pose embedding is zeroed)."""
    b, t = x.shape[:2]
    if t != cfg.max_tokens:
        raise AssertionError(f"temporal length must be {cfg.max_tokens}, got {t}")
    if external_cond is None:
        raise AssertionError("camera-pose conditioning is required")
    ps = cfg.patch_size
    h = F.conv2d(x.flatten(0, 1), p["embed_input.proj.weight"], p["embed_input.proj.bias"], stride=ps)

    pe = "external_cond_embedding.patch_embedder.proj."
    pose = F.conv2d(external_cond.flatten(0, 1), p[pe + "weight"], p[pe + "bias"], stride=ps)
    pose = pose.view(b, t, *pose.shape[1:])
    if external_cond_mask is not None:
        pose = torch.where(external_cond_mask.view(b, 1, 1, 1, 1), torch.zeros_like(pose), pose)
    emb = noise_level_embedding(p, noise_levels)[..., None, None] + pose
    emb = emb.flatten(0, 1)
    embs = [emb if l == 0 else F.avg_pool2d(emb, 2 ** l, 2 ** l) for l in range(cfg.num_levels)]
    if taps is not None:
        taps["emb0"] = emb

    angles = {}
    for lvl, kind in enumerate(cfg.block_types):
        if kind == "TransformerBlock":
            r = cfg.level_res(lvl)
            angles[lvl] = rope3d_angles(cfg.channels[lvl] // cfg.num_heads, (cfg.max_tokens, r, r), cfg.rope_theta).to(x.device)

    def ckpt(fn, *args):
        if CHECKPOINT_BLOCKS and torch.is_grad_enabled():
            from torch.utils.checkpoint import checkpoint
            return checkpoint(fn, *args, use_reentrant=False)
        return fn(*args)

    def run_level(h, lvl, prefixes):
        if cfg.block_types[lvl] == "ResBlock":
            for pre in prefixes:
                h = ckpt(lambda hh_, pre=pre: res_block(p, pre, hh_, embs[lvl], cfg), h)
            return h
        hh, ww = h.shape[-2:]
        tok = h.view(b, t, -1, hh, ww).permute(0, 1, 3, 4, 2).reshape(b, t * hh * ww, -1)
        etok = embs[lvl].view(b, t, -1, hh, ww).permute(0, 1, 3, 4, 2).reshape(b, t * hh * ww, -1)
        for pre in prefixes:
            tok = ckpt(lambda tk_, pre=pre: transformer_block(p, pre, tk_, etok, angles[lvl], cfg), tok)
        return tok.view(b, t, hh, ww, -1).permute(0, 1, 4, 2, 3).reshape(b * t, -1, hh, ww)

    before, after = [], []
    for lvl, n in enumerate(cfg.num_updown_blocks):
        h = run_level(h, lvl, [f"down_blocks.{lvl}.{i}" for i in range(n)])
        before.append(h)
        h = F.conv2d(F.avg_pool2d(h, 2, 2), p[f"down_blocks.{lvl}.{n}.conv.weight"],
                     p[f"down_blocks.{lvl}.{n}.conv.bias"], padding=1)
        after.append(h)
        if taps is not None:
            taps[f"down{lvl}"] = h
    top = cfg.num_levels - 1
    h = run_level(h, top, [f"mid_blocks.{i}" for i in range(cfg.num_mid_blocks)])
    if taps is not None:
        taps["mid"] = h
    for j, lvl in enumerate(reversed(range(top))):
        h = h - after.pop()
        h = F.conv2d(h, p[f"up_blocks.{j}.0.conv.weight"], p[f"up_blocks.{j}.0.conv.bias"], padding=1)
        h = F.interpolate(h, scale_factor=2, mode="nearest") + before.pop()
        h = run_level(h, lvl, [f"up_blocks.{j}.{i + 1}" for i in range(cfg.num_updown_blocks[lvl])])
        if taps is not None:
            taps[f"up{lvl}"] = h
    out = F.conv_transpose2d(h, p["project_output.proj.weight"], p["project_output.proj.bias"], stride=ps)
    return out.view(b, t, *out.shape[1:])
