"""Camera pose -> per-pixel ray encoding (oracle, CPU fp32).

Restates:
  * CameraPose.from_vectors / normalize_by_first / rays -- utils/geometry_utils.py:102-133,244-295
  * Ray.to_pos_encoding / _nerf_pos_encoding           -- utils/geometry_utils.py:41-81
  * DFoTVideoPose._process_conditions (ray_encoding)   -- algorithms/dfot/dfot_video_pose.py:64-110

Raw pose vector layout (16 floats per frame): [fx, fy, px, py | row-major 3x4 (R|T)].
Output channel order (180): origin then direction; within each: [sin block (3 comps x 15
freqs, component-major) | cos block (same order)], cos realised as sin(arg + pi/2).
"""
from __future__ import annotations

import math

import torch

N_FREQ = 15


def split_pose(raw: torch.Tensor):
    k = raw[..., :4]
    rt = raw[..., 4:].reshape(*raw.shape[:-1], 3, 4)
    return k, rt[..., :3], rt[..., 3]


def relative_to_first(rot: torch.Tensor, trans: torch.Tensor):
    """R' = R R0^T ; T' = T - R' T0   (frame 0 becomes the world frame)."""
    r0t = rot[:, 0].transpose(-1, -2)
    rot_rel = torch.einsum("btij,bjk->btik", rot, r0t)
    trans_rel = trans - torch.einsum("btij,bj->bti", rot_rel, trans[:, 0])
    return rot_rel, trans_rel


def rays(intr: torch.Tensor, rot: torch.Tensor, trans: torch.Tensor, resolution: int):
    """origin (B,T,H,W,3), direction (B,T,H,W,3) in world coordinates."""
    grid = torch.linspace(0, resolution - 1, resolution, dtype=intr.dtype) + 0.5
    u = grid.view(1, 1, 1, resolution)  # varies along W
    v = grid.view(1, 1, resolution, 1)  # varies along H
    scaled = intr * resolution
    fx, fy, px, py = (scaled[..., i].view(*intr.shape[:2], 1, 1) for i in range(4))
    x = ((u - px) / fx).expand(-1, -1, resolution, -1)
    y = ((v - py) / fy).expand(-1, -1, -1, resolution)
    cam_dir = torch.stack([x, y, torch.ones_like(x)], dim=-1)
    rot_inv = rot.transpose(-1, -2)
    direction = torch.einsum("btij,bthwj->bthwi", rot_inv, cam_dir)
    origin = -torch.einsum("btij,btj->bti", rot_inv, trans)
    origin = origin[:, :, None, None, :].expand(-1, -1, resolution, resolution, -1)
    return origin, direction


def nerf_encoding(x: torch.Tensor, n_freq: int = N_FREQ) -> torch.Tensor:
    scale = 2 ** torch.linspace(0, n_freq - 1, n_freq, dtype=x.dtype) * math.pi
    arg = (x[..., None] * scale).flatten(-2)  # (..., 3*n_freq) component-major
    return torch.sin(torch.cat([arg, arg + 0.5 * math.pi], dim=-1))


def ray_encoding(raw_poses: torch.Tensor, resolution: int) -> torch.Tensor:
    """(B,T,16) raw poses -> (B,T,180,H,W) fp32 conditioning tensor."""
    raw = raw_poses.to(torch.float32)
    intr, rot, trans = split_pose(raw)
    rot, trans = relative_to_first(rot, trans)
    origin, direction = rays(intr, rot, trans, resolution)
    enc = torch.cat([nerf_encoding(origin), nerf_encoding(direction)], dim=-1)
    return enc.permute(0, 1, 4, 2, 3).contiguous()
