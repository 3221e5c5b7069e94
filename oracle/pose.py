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


# --------------------------------------------------------------------------------------------------------------------
# Pose options off the BASELINE configs: normalize_by "mean", `bound`, and the interpolation of masked poses the reference applies
# under `temporal` History Guidance (utils/geometry_utils.py:135-205, dfot_video_pose.py:75-95).
# PARITY UNPINNED: these reference functions call roma==1.5.2.1 (rotmat_to_unitquat / unitquat_to_rotmat / unitquat_slerp), which is
# not importable here and not vendored, and the reference's tests hold no vector for them.  Restated from roma's published
# algorithm: quaternions are (x, y, z, w); rotmat_to_unitquat is the SciPy branch-on-largest-diagonal construction (no sign
# canonicalisation); unitquat_to_rotmat is the homogeneous quadratic form (so a non-unit q yields |q|^2 R -- which is what the
# reference's normalize_by_mean feeds it: the plain mean of the quaternions); unitquat_slerp is q0 * exp(t log(q0^-1 q1)) along
# the shortest arc.
# --------------------------------------------------------------------------------------------------------------------
def rotmat_to_unitquat(rot: torch.Tensor) -> torch.Tensor:
    m = rot.reshape(-1, 3, 3)
    diag = torch.diagonal(m, dim1=1, dim2=2)
    dec = torch.cat([diag, diag.sum(1, keepdim=True)], 1)
    choice = dec.argmax(1)
    q = torch.empty(m.shape[0], 4, dtype=rot.dtype)
    for n in range(m.shape[0]):
        c = int(choice[n])
        if c != 3:
            i, j, k = c, (c + 1) % 3, (c + 2) % 3
            q[n, i] = 1 - dec[n, 3] + 2 * m[n, i, i]
            q[n, j] = m[n, j, i] + m[n, i, j]
            q[n, k] = m[n, k, i] + m[n, i, k]
            q[n, 3] = m[n, k, j] - m[n, j, k]
        else:
            q[n, 0] = m[n, 2, 1] - m[n, 1, 2]
            q[n, 1] = m[n, 0, 2] - m[n, 2, 0]
            q[n, 2] = m[n, 1, 0] - m[n, 0, 1]
            q[n, 3] = 1 + dec[n, 3]
    q = q / q.norm(dim=1, keepdim=True)
    return q.reshape(*rot.shape[:-2], 4)


def unitquat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    x, y, z, w = q.unbind(-1)
    rows = [x * x - y * y - z * z + w * w, 2 * (x * y - z * w), 2 * (x * z + y * w),
            2 * (x * y + z * w), -x * x + y * y - z * z + w * w, 2 * (y * z - x * w),
            2 * (x * z - y * w), 2 * (y * z + x * w), -x * x - y * y + z * z + w * w]
    return torch.stack(rows, -1).reshape(*q.shape[:-1], 3, 3)


def quat_product(p: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    pv, pw, qv, qw = p[..., :3], p[..., 3:], q[..., :3], q[..., 3:]
    return torch.cat([pw * qv + qw * pv + torch.cross(pv, qv, dim=-1), pw * qw - (pv * qv).sum(-1, keepdim=True)], -1)


def unitquat_slerp(q0: torch.Tensor, q1: torch.Tensor, steps: torch.Tensor) -> torch.Tensor:
    """(S,4): q0 * exp(steps * log(q0^-1 q1)), shortest arc"""
    rel = quat_product(torch.cat([-q0[:3], q0[3:]]), q1)
    if rel[3] < 0:
        rel = -rel
    nv = rel[:3].norm()
    angle = 2 * torch.atan2(nv, rel[3])
    axis = rel[:3] / nv if nv > 1e-12 else torch.zeros(3, dtype=q0.dtype)
    half = 0.5 * steps * angle
    rots = torch.cat([torch.sin(half)[:, None] * axis[None], torch.cos(half)[:, None]], 1)
    return quat_product(q0[None].expand(steps.shape[0], 4), rots)


def replace_with_interpolation(rot: torch.Tensor, trans: torch.Tensor, mask: torch.Tensor):
    """geometry_utils.py:163-205: masked poses become slerp / lerp of the nearest unmasked ones, ends are held; every rotation
    (masked or not) takes the round trip through quaternions as in the reference"""
    q = rotmat_to_unitquat(rot)
    t = trans.clone()
    for b in range(mask.shape[0]):
        mk = mask[b]
        if not mk.any() or mk.all():
            continue
        valid = torch.where(~mk)[0].tolist()
        q[b, : valid[0]], t[b, : valid[0]] = q[b, valid[0]], t[b, valid[0]]
        q[b, valid[-1] + 1:], t[b, valid[-1] + 1:] = q[b, valid[-1]], t[b, valid[-1]]
        for lo, hi in zip(valid[:-1], valid[1:]):
            if hi - lo == 1:
                continue
            w = torch.linspace(0, 1, hi - lo + 1)
            q[b, lo: hi + 1] = unitquat_slerp(q[b, lo].clone(), q[b, hi].clone(), w)
            t[b, lo: hi + 1] = torch.lerp(t[b, lo].clone(), t[b, hi].clone(), w[:, None])
    return unitquat_to_rotmat(q), t


def relative_to(rot: torch.Tensor, trans: torch.Tensor, r_ref: torch.Tensor, t_ref: torch.Tensor):
    rot_rel = torch.einsum("btij,bkj->btik", rot, r_ref)
    return rot_rel, trans - torch.einsum("btij,bj->bti", rot_rel, t_ref)


def relative_to_mean(rot: torch.Tensor, trans: torch.Tensor):
    """geometry_utils.py:135-151 (the mean quaternion is NOT renormalised there)"""
    r_mean = unitquat_to_rotmat(rotmat_to_unitquat(rot).mean(dim=1))
    t_world = torch.einsum("btji,btj->bti", rot, trans).mean(dim=1)
    return relative_to(rot, trans, r_mean, torch.einsum("bij,bj->bi", r_mean, t_world))


def normalized_poses(raw_poses: torch.Tensor, normalize_by: str = "first", bound=None, interpolate_mask: torch.Tensor | None = None):
    """(intrinsics, R, T) after the option handling of DFoTVideoPose._process_conditions (dfot_video_pose.py:75-98)"""
    raw = raw_poses.to(torch.float32)
    intr, rot, trans = split_pose(raw)
    if interpolate_mask is not None:
        rot, trans = replace_with_interpolation(rot, trans, interpolate_mask)
    if normalize_by == "first":
        rot, trans = relative_to_first(rot, trans)
    elif normalize_by == "mean":
        rot, trans = relative_to_mean(rot, trans)
    else:
        raise ValueError(f"Unknown camera pose normalization method: {normalize_by}")
    if bound is not None:  # scale_within_bounds :153-161 (per video and per axis)
        trans = trans * (bound / trans.abs().max(dim=1, keepdim=True).values.clamp(min=1e-6))
    return intr, rot, trans


def process_conditions(raw_poses: torch.Tensor, resolution: int, normalize_by: str = "first", bound=None,
                       interpolate_mask: torch.Tensor | None = None) -> torch.Tensor:
    """DFoTVideoPose._process_conditions (dfot_video_pose.py:64-110) with every option; ray_encoding() above is the default path."""
    intr, rot, trans = normalized_poses(raw_poses, normalize_by, bound, interpolate_mask)
    origin, direction = rays(intr, rot, trans, resolution)
    enc = torch.cat([nerf_encoding(origin), nerf_encoding(direction)], dim=-1)
    return enc.permute(0, 1, 4, 2, 3).contiguous()
