"""DiT3D backbone forward (oracle, CPU fp32, functional over a state dict) -- TEST INFRASTRUCTURE ONLY.

Restates the reference's Kinetics-600 backbone (BASELINE config 4):
  * DiT3D.forward / unpatchify                -- algorithms/dfot/backbones/dit/dit3d.py:146-192
    (timm==1.0.17 PatchEmbed: Conv2d k=s=patch, flatten -> "(b t) p c")
  * DiTBase "full" variant with rope_3d       -- dit/dit_base.py:150-196, 277-285, 391-419
  * DiTBlock (this fork's semantics: the AdaLN-Zero output REPLACES the stream, i.e. x <- m + gate*attn(m) with
    m = modulate(LN(x)); the MLP branch exists only when spatial_mlp_ratio > 0)   -- dit/dit_blocks.py:440-510
  * AdaLayerNorm / AdaLayerNormZero / modulate -- dit/dit_blocks.py:17-18, 378-437
  * Attention (qkv Linear+bias, RoPE on q,k, softmax(QK^T/sqrt(d))V, proj)       -- dit/dit_blocks.py:49-128
  * DITFinalLayer                              -- dit/dit_blocks.py:513-542
  * sinusoidal Timesteps (flip_sin_to_cos, shift 0) + TimestepEmbedding MLP      -- modules/embeddings.py:12-31, 67-93,
    113-155 (diffusers==0.32.2 TimestepEmbedding: linear_1 -> SiLU -> linear_2)
  * RotaryEmbedding3D                          -- modules/embeddings.py:158-277
No external condition (kinetics_600 has external_cond_dim 0); no causal mask (dit3d.py:23-26 rejects it).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .uvit import apply_rope, rope3d_angles

Params = Dict[str, torch.Tensor]


@dataclass
class DiTConfig:
    hidden_size: int = 1152
    depth: int = 28
    num_heads: int = 16
    patch_size: int = 1
    in_channels: int = 16
    resolution: Tuple[int, int] = (16, 16)
    max_tokens: int = 5
    spatial_mlp_ratio: float = 0.0  # dit3d.yaml leaves it unset -> attention-only blocks
    noise_dim: int = 256
    rope_theta: float = 10000.0
    eps: float = 1e-6

    @property
    def grid(self) -> Tuple[int, int]:
        return self.resolution[0] // self.patch_size, self.resolution[1] // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid[0] * self.grid[1]

    @property
    def out_channels(self) -> int:
        return self.patch_size ** 2 * self.in_channels

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def mlp_hidden(self) -> int:
        return int(self.hidden_size * self.spatial_mlp_ratio) if self.spatial_mlp_ratio else 0


def param_shapes(cfg: DiTConfig) -> Dict[str, tuple]:
    """Every persistent state-dict entry of the reference module, in its registration order."""
    h, nd, p = cfg.hidden_size, cfg.noise_dim, cfg.patch_size
    s: Dict[str, tuple] = {
        "noise_level_pos_embedding.embedding.linear_1.weight": (h, nd),
        "noise_level_pos_embedding.embedding.linear_1.bias": (h,),
        "noise_level_pos_embedding.embedding.linear_2.weight": (h, h),
        "noise_level_pos_embedding.embedding.linear_2.bias": (h,),
        "patch_embedder.proj.weight": (h, cfg.in_channels, p, p),
        "patch_embedder.proj.bias": (h,),
    }
    for i in range(cfg.depth):
        pre = f"dit_base.blocks.{i}"
        s[f"{pre}.norm1.modulation.1.weight"] = (3 * h, h)
        s[f"{pre}.norm1.modulation.1.bias"] = (3 * h,)
        s[f"{pre}.attn.qkv.weight"] = (3 * h, h)
        s[f"{pre}.attn.qkv.bias"] = (3 * h,)
        s[f"{pre}.attn.proj.weight"] = (h, h)
        s[f"{pre}.attn.proj.bias"] = (h,)
        if cfg.mlp_hidden:
            s[f"{pre}.norm2.modulation.1.weight"] = (3 * h, h)
            s[f"{pre}.norm2.modulation.1.bias"] = (3 * h,)
            s[f"{pre}.mlp.fc1.weight"] = (cfg.mlp_hidden, h)
            s[f"{pre}.mlp.fc1.bias"] = (cfg.mlp_hidden,)
            s[f"{pre}.mlp.fc2.weight"] = (h, cfg.mlp_hidden)
            s[f"{pre}.mlp.fc2.bias"] = (h,)
    s["dit_base.final_layer.norm_final.modulation.1.weight"] = (2 * h, h)
    s["dit_base.final_layer.norm_final.modulation.1.bias"] = (2 * h,)
    s["dit_base.final_layer.linear.weight"] = (cfg.out_channels, h)
    s["dit_base.final_layer.linear.bias"] = (cfg.out_channels,)
    return s


def seeded_params(cfg: DiTConfig, seed: int = 0) -> Params:
    """Deterministic non-degenerate weights.  The reference zero-inits every modulation and the final linear
    (dit_blocks.py:392-395, 422-425, 528-531), which would make the output identically zero and hide every block, so
    parity tests draw ALL tensors with fan-in scaling instead (modulations at a smaller gain so 28 blocks stay O(1))."""
    g = torch.Generator().manual_seed(seed)
    out: Params = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith(".bias"):
            out[name] = 0.05 * torch.randn(shape, generator=g)
            continue
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        gain = 0.5 if ".modulation." in name else 1.0
        out[name] = gain * torch.randn(shape, generator=g) / math.sqrt(fan_in)
    return out


def timestep_features(k: torch.Tensor, dim: int) -> torch.Tensor:
    """get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin] of k * 10000^(-i/half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = k[..., None].float() * freqs
    return torch.cat([a.cos(), a.sin()], dim=-1)


def noise_level_embedding(p: Params, cfg: DiTConfig, k: torch.Tensor) -> torch.Tensor:
    pre = "noise_level_pos_embedding.embedding"
    f = timestep_features(k, cfg.noise_dim)
    return F.linear(F.silu(F.linear(f, p[f"{pre}.linear_1.weight"], p[f"{pre}.linear_1.bias"])),
                    p[f"{pre}.linear_2.weight"], p[f"{pre}.linear_2.bias"])


def _layer_norm(x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.layer_norm(x, x.shape[-1:], None, None, eps)


def _ada_ln(p: Params, pre: str, x: torch.Tensor, c: torch.Tensor, chunks: int, eps: float):
    mod = F.linear(F.silu(c), p[f"{pre}.modulation.1.weight"], p[f"{pre}.modulation.1.bias"]).chunk(chunks, dim=-1)
    m = _layer_norm(x, eps) * (1 + mod[1]) + mod[0]
    return (m, mod[2]) if chunks == 3 else m


def attention(p: Params, pre: str, x: torch.Tensor, ang: torch.Tensor, heads: int) -> torch.Tensor:
    b, n, c = x.shape
    d = c // heads
    qkv = F.linear(x, p[f"{pre}.qkv.weight"], p[f"{pre}.qkv.bias"]).reshape(b, n, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    q, k = apply_rope(q, ang[:n]), apply_rope(k, ang[:n])
    w = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(d), dim=-1)
    o = (w @ v).transpose(1, 2).reshape(b, n, c)
    return F.linear(o, p[f"{pre}.proj.weight"], p[f"{pre}.proj.bias"])


def dit_block(p: Params, pre: str, x: torch.Tensor, c: torch.Tensor, ang: torch.Tensor, cfg: DiTConfig) -> torch.Tensor:
    m, gate = _ada_ln(p, f"{pre}.norm1", x, c, 3, cfg.eps)
    x = m + gate * attention(p, f"{pre}.attn", m, ang, cfg.num_heads)
    if cfg.mlp_hidden:
        m, gate = _ada_ln(p, f"{pre}.norm2", x, c, 3, cfg.eps)
        hid = F.gelu(F.linear(m, p[f"{pre}.mlp.fc1.weight"], p[f"{pre}.mlp.fc1.bias"]), approximate="tanh")
        x = m + gate * F.linear(hid, p[f"{pre}.mlp.fc2.weight"], p[f"{pre}.mlp.fc2.bias"])
    return x


def forward(p: Params, cfg: DiTConfig, x: torch.Tensor, noise_levels: torch.Tensor,
            taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """x [B,T,C,H,W] fp32, noise_levels [B,T] (integer level indices, any dtype) -> [B,T,C,H,W]."""
    b, t, ch, hh, ww = x.shape
    ps, (gh, gw), h = cfg.patch_size, cfg.grid, cfg.hidden_size
    tok = F.conv2d(x.reshape(b * t, ch, hh, ww), p["patch_embedder.proj.weight"], p["patch_embedder.proj.bias"], stride=ps)
    tok = tok.flatten(2).transpose(1, 2).reshape(b, t * gh * gw, h)
    emb = noise_level_embedding(p, cfg, noise_levels)  # [B,T,h]
    c = emb[:, :, None, :].expand(b, t, gh * gw, h).reshape(b, t * gh * gw, h)
    ang = rope3d_angles(cfg.head_dim, (cfg.max_tokens, gh, gw), cfg.rope_theta)
    if taps is not None:
        taps["emb"], taps["tokens"] = emb, tok
    for i in range(cfg.depth):
        tok = dit_block(p, f"dit_base.blocks.{i}", tok, c, ang, cfg)
        if taps is not None:
            taps[f"block{i}"] = tok
    tok = _ada_ln(p, "dit_base.final_layer.norm_final", tok, c, 2, cfg.eps)
    out = F.linear(tok, p["dit_base.final_layer.linear.weight"], p["dit_base.final_layer.linear.bias"])
    out = out.reshape(b * t, gh, gw, ps, ps, ch).permute(0, 1, 3, 2, 4, 5).reshape(b * t, gh * ps, gw * ps, ch)
    return out.permute(0, 3, 1, 2).reshape(b, t, ch, hh, ww)
