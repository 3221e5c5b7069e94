"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain PyTorch, fp32) of the reference's VideoVAE decode path, functional over a state
dict with the reference's key names.  Never imported by the product (tests/test_layout.py).

Follows:
  VideoVAE.decode / _decode          algorithms/vae/video_vae/model.py:445-476   (post_quant_conv, Decoder, last `desired_length` frames)
  Decoder.forward                    algorithms/vae/video_vae/model.py:255-281
  PaddedConv3D (causal)              algorithms/vae/common/modules/conv.py:40-114   first frame repeated kt-1 times in front, no temporal padding
  ResnetBlock3D                      algorithms/vae/common/modules/resnet.py:62-109
  AttnBlock3D                        algorithms/vae/common/modules/attention.py:96-156  per-frame attention over H*W with all C channels
  SpatialUpsample2x                  algorithms/vae/common/modules/updownsample.py:53-83
  Spatial2xTime2x3DUpsample (causal) algorithms/vae/common/modules/updownsample.py:121-156
  Normalize = GroupNorm(32, eps 1e-6), nonlinearity = x * sigmoid(x)   normalize.py, ops.py
  BaseVideoAlgo._decode / _run_vae   algorithms/common/base_pytorch_video_algo.py:553-629
Pinned by tests/golden/vae_decode.npz, captured from the reference's own VideoVAE source (tools/make_golden.py, ONLY=vae).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


@dataclass
class VAEConfig:
    z_channels: int = 16
    hidden_size: int = 128
    hidden_size_mult: Sequence[int] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    embed_dim: int = 16
    use_quant_layer: bool = True
    spatial_upsample: Sequence[str] = ("", "SpatialUpsample2x", "Spatial2xTime2x3DUpsample", "Spatial2xTime2x3DUpsample")


def causal_conv3d(p: Params, name: str, x: torch.Tensor, padding: int) -> torch.Tensor:
    w, b = p[name + ".conv.weight"], p[name + ".conv.bias"]
    kt = w.shape[2]
    if kt > 1:
        x = torch.cat([x[:, :, :1].repeat(1, 1, kt - 1, 1, 1), x], dim=2)
    return F.conv3d(x, w, b, padding=(0, padding, padding))


def group_norm(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.group_norm(x, 32, p[name + ".weight"], p[name + ".bias"], eps=1e-6)


def silu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(x)


def resnet_block3d(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    h = causal_conv3d(p, name + ".conv1", silu(group_norm(p, name + ".norm1", x)), 1)
    h = causal_conv3d(p, name + ".conv2", silu(group_norm(p, name + ".norm2", h)), 1)
    if name + ".nin_shortcut.conv.weight" in p:
        x = causal_conv3d(p, name + ".nin_shortcut", x, 0)
    return x + h


def attn_block3d(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    h = group_norm(p, name + ".norm", x)
    q, k, v = (causal_conv3d(p, f"{name}.{n}", h, 0) for n in ("q", "k", "v"))
    b, c, t, hh, ww = q.shape
    fr = lambda a: a.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh * ww)
    q, k, v = fr(q).permute(0, 2, 1), fr(k), fr(v)
    w = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)
    o = torch.bmm(v, w.permute(0, 2, 1)).reshape(b, t, c, hh, ww).permute(0, 2, 1, 3, 4)
    return x + causal_conv3d(p, name + ".proj_out", o, 0)


def upsample(p: Params, name: str, kind: str, x: torch.Tensor) -> torch.Tensor:
    if kind == "SpatialUpsample2x":
        b, c, t, h, w = x.shape
        x = F.interpolate(x.reshape(b, c * t, h, w), scale_factor=(2, 2), mode="nearest").reshape(b, c, t, 2 * h, 2 * w)
        return causal_conv3d(p, name + ".conv", x, 1)
    if kind == "Spatial2xTime2x3DUpsample":
        if x.size(2) > 1:
            first, rest = x[:, :, :1], x[:, :, 1:]
            rest = F.interpolate(rest, scale_factor=(2, 2, 2), mode="trilinear")
            first = F.interpolate(first, scale_factor=(1, 2, 2), mode="trilinear")
            x = torch.cat([first, rest], dim=2)
        else:
            x = F.interpolate(x, scale_factor=(1, 2, 2), mode="trilinear")
        return causal_conv3d(p, name + ".conv", x, 1)
    raise ValueError(f"oracle: unsupported upsample {kind}")


def decode(p: Params, cfg: VAEConfig, z: torch.Tensor, desired_length: Optional[int] = None) -> torch.Tensor:
    """z (B, embed_dim, T, H, W) -> (B, 3, T', H', W')"""
    if cfg.use_quant_layer:
        z = causal_conv3d(p, "post_quant_conv", z, 0)
    h = causal_conv3d(p, "decoder.conv_in", z, 1)
    h = resnet_block3d(p, "decoder.mid.block_1", h)
    h = attn_block3d(p, "decoder.mid.attn_1", h)
    h = resnet_block3d(p, "decoder.mid.block_2", h)
    for lvl in reversed(range(len(cfg.hidden_size_mult))):
        for i in range(cfg.num_res_blocks + 1):
            h = resnet_block3d(p, f"decoder.up.{lvl}.block.{i}", h)
        if cfg.spatial_upsample[lvl]:
            h = upsample(p, f"decoder.up.{lvl}.upsample", cfg.spatial_upsample[lvl], h)
    h = causal_conv3d(p, "decoder.conv_out", silu(group_norm(p, "decoder.norm_out", h)), 1)
    if desired_length is not None:
        h = h[:, :, -desired_length:]
        assert h.shape[2] == desired_length
    return h


def decode_latents(p: Params, cfg: VAEConfig, latents: torch.Tensor, n_frames: int, vae_batch_size: int = 2) -> torch.Tensor:
    """_decode for a VideoVAE: latents (b t c h w) -> frames (b t c h w) in [0, 1] ("* 0.5 + 0.5"), chunks of vae_batch_size videos"""
    x = latents.permute(0, 2, 1, 3, 4)
    n = (x.shape[0] + vae_batch_size - 1) // vae_batch_size
    outs = [decode(p, cfg, ch, n_frames) * 0.5 + 0.5 for ch in torch.chunk(x, n, 0)]
    return torch.cat(outs, 0).permute(0, 2, 1, 3, 4)


def seeded_tensor(name: str, shape, seed: int = 71) -> torch.Tensor:
    """order-independent seeded weights (one generator per tensor, seeded by the key name): the recipe the golden fixture was made with"""
    import zlib
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 1000003 * seed) % (2 ** 31))
    shape = tuple(shape)
    if name.endswith("bias"):
        return 0.02 * torch.randn(shape, generator=g)
    if len(shape) == 1:
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return torch.randn(shape, generator=g) / (fan_in ** 0.5)
