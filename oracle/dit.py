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
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half).to(k.device)
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
    ang = rope3d_angles(cfg.head_dim, (cfg.max_tokens, gh, gw), cfg.rope_theta).to(x.device)
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


# ============================================================================================================
# DifferenceDiT3D, variant "factorized_matrix_attention" (the bash/k600 model)
#   * DifferenceDiT3D.forward / create_diff_index        -- dit/difference_dit3d.py:159-226
#   * DiTBase factorized-matrix wiring, sinusoidal_2d     -- dit/dit_base.py:155-226, 247-251, 352-419
#   * MatrixAttention / matrix_mul                        -- dit/dit_blocks.py:211-350
#   * MatrixDiTBlock                                      -- dit/dit_blocks.py:549-652
#   * SinusoidalPositionalEmbedding / get_nd_sincos_pos_embed -- dit/dit_base.py:505-572
#   (diffusers==0.32.2 LabelEmbedding with dropout 0: a plain nn.Embedding(2, hidden) named embedding_table;
#    timm==1.0.17 Mlp: fc1 -> GELU(tanh) -> fc2)
# ============================================================================================================
@dataclass
class DiffDiTConfig:
    hidden_size: int = 1152        # embed_row_dim (FacMatDiT/group_XL/XL-64-1.yaml)
    depth: int = 28
    num_heads: int = 12            # spatial heads (difference_dit3d_factorized_matrix.yaml)
    patch_size: int = 1
    in_channels: int = 16
    resolution: Tuple[int, int] = (16, 16)
    max_tokens: int = 5            # the backbone sees 2 * max_tokens merged (difference, frame) tokens
    embed_col_dim: int = 64
    num_col_heads: int = 1
    num_row_heads: int = 16
    mlp_ratio: float = 4.0         # temporal (matrix) blocks
    spatial_mlp_ratio: float = 4.0
    use_bias: bool = True
    noise_dim: int = 256
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


def diff_param_shapes(cfg: DiffDiTConfig) -> Dict[str, tuple]:
    h, nd, p, pn, e = cfg.hidden_size, cfg.noise_dim, cfg.patch_size, cfg.num_patches, cfg.embed_col_dim
    s: Dict[str, tuple] = {
        "noise_level_pos_embedding.embedding.linear_1.weight": (h, nd),
        "noise_level_pos_embedding.embedding.linear_1.bias": (h,),
        "noise_level_pos_embedding.embedding.linear_2.weight": (h, h),
        "noise_level_pos_embedding.embedding.linear_2.bias": (h,),
        "patch_embedder.proj.weight": (h, cfg.in_channels, p, p),
        "patch_embedder.proj.bias": (h,),
        "diff_embedder.embedding_table.weight": (2, h),
    }

    def mlp(pre: str, ratio: float):
        hid = int(h * ratio)
        s[f"{pre}.norm2.modulation.1.weight"] = (3 * h, h)
        s[f"{pre}.norm2.modulation.1.bias"] = (3 * h,)
        s[f"{pre}.mlp.fc1.weight"] = (hid, h)
        s[f"{pre}.mlp.fc1.bias"] = (hid,)
        s[f"{pre}.mlp.fc2.weight"] = (h, hid)
        s[f"{pre}.mlp.fc2.bias"] = (h,)

    for i in range(cfg.depth):
        pre = f"dit_base.blocks.{i}"
        s[f"{pre}.norm1.modulation.1.weight"] = (3 * h, h)
        s[f"{pre}.norm1.modulation.1.bias"] = (3 * h,)
        s[f"{pre}.attn.qkv.weight"] = (3 * h, h)
        s[f"{pre}.attn.qkv.bias"] = (3 * h,)
        s[f"{pre}.attn.proj.weight"] = (h, h)
        s[f"{pre}.attn.proj.bias"] = (h,)
        if cfg.spatial_mlp_ratio:
            mlp(pre, cfg.spatial_mlp_ratio)
    for i in range(cfg.depth):
        pre = f"dit_base.temporal_blocks.{i}"
        s[f"{pre}.norm1.modulation.1.weight"] = (3 * h, h)
        s[f"{pre}.norm1.modulation.1.bias"] = (3 * h,)
        s[f"{pre}.attn.qkv_u"] = (pn, e)
        s[f"{pre}.attn.proj_u"] = (e, pn)
        s[f"{pre}.attn.qkv_v"] = (h, 3 * h)
        s[f"{pre}.attn.proj_v"] = (h, h)
        if cfg.use_bias:
            s[f"{pre}.attn.qkv_bias"] = (e, 3 * h)
            s[f"{pre}.attn.proj_bias"] = (pn, h)
        if cfg.mlp_ratio:
            mlp(pre, cfg.mlp_ratio)
    s["dit_base.final_layer.norm_final.modulation.1.weight"] = (2 * h, h)
    s["dit_base.final_layer.norm_final.modulation.1.bias"] = (2 * h,)
    s["dit_base.final_layer.linear.weight"] = (cfg.out_channels, h)
    s["dit_base.final_layer.linear.bias"] = (cfg.out_channels,)
    return s


def diff_seeded_params(cfg: DiffDiTConfig, seed: int = 0) -> Params:
    """Non-degenerate weights (see seeded_params).  The matrix factors are (in, out) matrices: fan-in = shape[0]."""
    g = torch.Generator().manual_seed(seed)
    out: Params = {}
    for name, shape in diff_param_shapes(cfg).items():
        leaf = name.rsplit(".", 1)[-1]
        if leaf in ("bias", "qkv_bias", "proj_bias"):
            out[name] = 0.05 * torch.randn(shape, generator=g)
        elif leaf in ("qkv_u", "proj_u", "qkv_v", "proj_v"):
            out[name] = torch.randn(shape, generator=g) / math.sqrt(shape[0])
        elif name.startswith("diff_embedder"):
            out[name] = 0.3 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            gain = 0.5 if ".modulation." in name else 1.0
            out[name] = gain * torch.randn(shape, generator=g) / math.sqrt(fan_in)
    return out


def sincos_2d(embed_dim: int, grid: Tuple[int, int]) -> torch.Tensor:
    """get_nd_sincos_pos_embed for a 2-D grid: np.meshgrid's default 'xy' indexing makes flattened entry m use
    position m % grid[0] for the first half of the channels and m // grid[0] for the second ([sin | cos] each)."""
    import numpy as np
    half = embed_dim // 2
    omega = 1.0 / 10000 ** (np.arange(half // 2, dtype=np.float64) / (half / 2.0))
    m = np.arange(grid[0] * grid[1])
    parts = []
    for pos in (m % grid[0], m // grid[0]):
        ang = np.einsum("m,d->md", pos.astype(np.float32), omega)
        parts.append(np.concatenate([np.sin(ang), np.cos(ang)], axis=1))
    return torch.from_numpy(np.concatenate(parts, axis=1)).float()


def _mlp_branch(p: Params, pre: str, x: torch.Tensor, c: torch.Tensor, eps: float) -> torch.Tensor:
    m, gate = _ada_ln(p, f"{pre}.norm2", x, c, 3, eps)
    hid = F.gelu(F.linear(m, p[f"{pre}.mlp.fc1.weight"], p[f"{pre}.mlp.fc1.bias"]), approximate="tanh")
    return m + gate * F.linear(hid, p[f"{pre}.mlp.fc2.weight"], p[f"{pre}.mlp.fc2.bias"])


def spatial_block(p: Params, pre: str, x: torch.Tensor, c: torch.Tensor, cfg: DiffDiTConfig) -> torch.Tensor:
    """DiTBlock on (B*T, P, C): attention inside one frame, no RoPE (sinusoidal_2d was added once)."""
    m, gate = _ada_ln(p, f"{pre}.norm1", x, c, 3, cfg.eps)
    b, n, ch = m.shape
    d = ch // cfg.num_heads
    qkv = F.linear(m, p[f"{pre}.attn.qkv.weight"], p[f"{pre}.attn.qkv.bias"]).reshape(b, n, 3, cfg.num_heads, d).permute(2, 0, 3, 1, 4)
    w = torch.softmax(qkv[0] @ qkv[1].transpose(-2, -1) / math.sqrt(d), dim=-1)
    o = (w @ qkv[2]).transpose(1, 2).reshape(b, n, ch)
    x = m + gate * F.linear(o, p[f"{pre}.attn.proj.weight"], p[f"{pre}.attn.proj.bias"])
    return _mlp_branch(p, pre, x, c, cfg.eps) if cfg.spatial_mlp_ratio else x


def matrix_attention(p: Params, pre: str, x: torch.Tensor, cfg: DiffDiTConfig) -> torch.Tensor:
    """x (B, L, P, C): every frame is ONE token, a P x C matrix projected by left/right factors (dit_blocks.py:211-350)."""
    b, l, _, _ = x.shape
    cc, rr = cfg.num_col_heads, cfg.num_row_heads
    hn, hd = cfg.embed_col_dim // cc, cfg.hidden_size // rr
    qkv = torch.einsum("nm,blnd,dk->blmk", p[f"{pre}.qkv_u"], x, p[f"{pre}.qkv_v"])
    if cfg.use_bias:
        qkv = qkv + p[f"{pre}.qkv_bias"]
    qkv = qkv.reshape(b, l, cc, hn, 3, rr, hd).permute(4, 0, 2, 5, 1, 3, 6)  # k b c r l n d
    q, k, v = qkv[0] * (hn * hd) ** -0.5, qkv[1], qkv[2]
    w = torch.softmax(torch.einsum("bcrlnd,bcrknd->bcrlk", q, k), dim=-1)
    o = torch.einsum("bcrlk,bcrknd->bcrlnd", w, v).permute(0, 3, 1, 4, 2, 5).reshape(b, l, cc * hn, rr * hd)
    o = torch.einsum("nm,blnd,dk->blmk", p[f"{pre}.proj_u"], o, p[f"{pre}.proj_v"])
    return o + p[f"{pre}.proj_bias"] if cfg.use_bias else o


def temporal_block(p: Params, pre: str, x: torch.Tensor, c: torch.Tensor, frames: int, cfg: DiffDiTConfig) -> torch.Tensor:
    """MatrixDiTBlock on (B, T*P, C)."""
    m, gate = _ada_ln(p, f"{pre}.norm1", x, c, 3, cfg.eps)
    b, n, ch = m.shape
    a = matrix_attention(p, f"{pre}.attn", m.reshape(b, frames, n // frames, ch), cfg).reshape(b, n, ch)
    x = m + gate * a
    return _mlp_branch(p, pre, x, c, cfg.eps) if cfg.mlp_ratio else x


def diff_forward(p: Params, cfg: DiffDiTConfig, x: torch.Tensor, noise_levels: torch.Tensor,
                 taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """x [B, 2T, C, H, W] interleaved (difference_0, frame_0, difference_1, ...), noise_levels [B, 2T] -> same shape as x."""
    b, t, ch, hh, ww = x.shape
    ps, (gh, gw), h = cfg.patch_size, cfg.grid, cfg.hidden_size
    pn = gh * gw
    idx = torch.tensor([1, 0] * (t // 2))  # create_diff_index(diff_first=True), merge_type "interleaved"
    emb = p["diff_embedder.embedding_table.weight"][idx][None] + noise_level_embedding(p, cfg, noise_levels)  # [B,2T,h]
    tok = F.conv2d(x.reshape(b * t, ch, hh, ww), p["patch_embedder.proj.weight"], p["patch_embedder.proj.bias"], stride=ps)
    tok = tok.flatten(2).transpose(1, 2) + sincos_2d(h, (gh, gw))[None]  # (B*T, P, h)
    c = emb.reshape(b * t, 1, h).expand(b * t, pn, h)
    for i in range(cfg.depth):
        tok = spatial_block(p, f"dit_base.blocks.{i}", tok, c, cfg)
        tok = temporal_block(p, f"dit_base.temporal_blocks.{i}", tok.reshape(b, t * pn, h), c.reshape(b, t * pn, h), t, cfg)
        tok = tok.reshape(b * t, pn, h)
        if taps is not None:
            taps[f"block{i}"] = tok.reshape(b, t * pn, h)
    tok = _ada_ln(p, "dit_base.final_layer.norm_final", tok, c, 2, cfg.eps)
    out = F.linear(tok, p["dit_base.final_layer.linear.weight"], p["dit_base.final_layer.linear.bias"])
    out = out.reshape(b * t, gh, gw, ps, ps, ch).permute(0, 1, 3, 2, 4, 5).reshape(b * t, gh * ps, gw * ps, ch)
    return out.permute(0, 3, 1, 2).reshape(b, t, ch, hh, ww)
