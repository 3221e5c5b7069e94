"""Host side of the camera-pose front end for the options off the default path (tiny: B*T poses of 16 floats, numpy fp32).

``DFoTVideoPose._process_conditions`` (algorithms/dfot/dfot_video_pose.py:64-110) normalises the poses by the first frame
(done inside ``dfot_ray_encode``) -- or, by configuration, by the mean frame (``normalize_by: mean``), rescales the camera
positions into ``[-bound, bound]^3``, and under ``temporal`` History Guidance first replaces the poses of fully masked frames
by an interpolation of the nearest unmasked ones (utils/geometry_utils.py:135-205).  Those cases are computed here and handed to
``dfot_ray_encode_normalized``.

The reference does the quaternion work with roma==1.5.2.1, which is not available offline: PARITY UNPINNED against the
reference; the conventions follow roma's published algorithm ((x,y,z,w) quaternions, SciPy's matrix->quaternion construction
without sign canonicalisation, the homogeneous quaternion->matrix form that the un-normalised mean quaternion of
``normalize_by_mean`` is pushed through, shortest-arc slerp) and are checked against SciPy's Rotation / Slerp in the tests.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

F = np.float32


def _to_quat(rot: np.ndarray) -> np.ndarray:
    m = rot.reshape(-1, 3, 3).astype(F)
    d = np.concatenate([np.diagonal(m, axis1=1, axis2=2), np.trace(m, axis1=1, axis2=2)[:, None]], 1).astype(F)
    q = np.empty((m.shape[0], 4), F)
    for n, c in enumerate(d.argmax(1)):
        if c != 3:
            i, j, k = c, (c + 1) % 3, (c + 2) % 3
            q[n, i] = F(1) - d[n, 3] + F(2) * m[n, i, i]
            q[n, j] = m[n, j, i] + m[n, i, j]
            q[n, k] = m[n, k, i] + m[n, i, k]
            q[n, 3] = m[n, k, j] - m[n, j, k]
        else:
            q[n, :3] = (m[n, 2, 1] - m[n, 1, 2], m[n, 0, 2] - m[n, 2, 0], m[n, 1, 0] - m[n, 0, 1])
            q[n, 3] = F(1) + d[n, 3]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.reshape(*rot.shape[:-2], 4)


def _to_rotmat(q: np.ndarray) -> np.ndarray:
    x, y, z, w = (q[..., i] for i in range(4))
    out = np.stack([x * x - y * y - z * z + w * w, 2 * (x * y - z * w), 2 * (x * z + y * w),
                    2 * (x * y + z * w), -x * x + y * y - z * z + w * w, 2 * (y * z - x * w),
                    2 * (x * z - y * w), 2 * (y * z + x * w), -x * x - y * y + z * z + w * w], -1)
    return out.reshape(*q.shape[:-1], 3, 3).astype(F)


def _qmul(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    pv, pw, qv, qw = p[..., :3], p[..., 3:], q[..., :3], q[..., 3:]
    return np.concatenate([pw * qv + qw * pv + np.cross(pv, qv), pw * qw - (pv * qv).sum(-1, keepdims=True)], -1).astype(F)


def _slerp(q0: np.ndarray, q1: np.ndarray, steps: np.ndarray) -> np.ndarray:
    rel = _qmul(np.concatenate([-q0[:3], q0[3:]]), q1)
    if rel[3] < 0:
        rel = -rel
    nv = np.linalg.norm(rel[:3])
    half = (F(0.5) * steps * F(2) * np.arctan2(nv, rel[3])).astype(F)
    axis = rel[:3] / nv if nv > 1e-12 else np.zeros(3, F)
    rots = np.concatenate([np.sin(half)[:, None] * axis[None], np.cos(half)[:, None]], 1).astype(F)
    return _qmul(np.broadcast_to(q0, (len(steps), 4)), rots)


def interpolate_masked(rot: np.ndarray, trans: np.ndarray, mask: np.ndarray):
    """CameraPose.replace_with_interpolation (geometry_utils.py:163-205)"""
    q, t = _to_quat(rot), trans.astype(F).copy()
    for b in range(mask.shape[0]):
        mk = mask[b]
        if not mk.any() or mk.all():
            continue
        valid = np.flatnonzero(~mk)
        q[b, : valid[0]], t[b, : valid[0]] = q[b, valid[0]], t[b, valid[0]]
        q[b, valid[-1] + 1:], t[b, valid[-1] + 1:] = q[b, valid[-1]], t[b, valid[-1]]
        for lo, hi in zip(valid[:-1], valid[1:]):
            if hi - lo == 1:
                continue
            w = np.linspace(0, 1, hi - lo + 1, dtype=F)
            q[b, lo: hi + 1] = _slerp(q[b, lo].copy(), q[b, hi].copy(), w)
            t[b, lo: hi + 1] = t[b, lo] + w[:, None] * (t[b, hi] - t[b, lo])
    return _to_rotmat(q), t


def normalize_poses(raw_poses: np.ndarray, normalize_by: str = "first", bound: Optional[float] = None,
                    interpolate_mask: Optional[np.ndarray] = None) -> np.ndarray:
    """(B,T,16) raw poses -> (B,T,16) poses in the world frame the reference's options select, for ``dfot_ray_encode_normalized``"""
    raw = np.asarray(raw_poses, F)
    if raw.ndim != 3 or raw.shape[-1] != 16:
        raise ValueError(f"raw camera poses must be (B, T, 16), got {raw.shape}")
    rt = raw[..., 4:].reshape(*raw.shape[:2], 3, 4)
    rot, trans = rt[..., :3].copy(), rt[..., 3].copy()
    if interpolate_mask is not None:
        rot, trans = interpolate_masked(rot, trans, np.asarray(interpolate_mask, bool))
    if normalize_by == "first":
        r_ref, t_ref = rot[:, 0], trans[:, 0]
    elif normalize_by == "mean":
        r_ref = _to_rotmat(_to_quat(rot).mean(axis=1, dtype=F))
        t_ref = np.einsum("bij,bj->bi", r_ref, np.einsum("btji,btj->bti", rot, trans).mean(axis=1, dtype=F))
    else:
        raise ValueError(f"Unknown camera pose normalization method: {normalize_by}")
    rot = np.einsum("btij,bkj->btik", rot, r_ref).astype(F)
    trans = (trans - np.einsum("btij,bj->bti", rot, t_ref)).astype(F)
    if bound is not None:
        trans = trans * (F(bound) / np.maximum(np.abs(trans).max(axis=1, keepdims=True), F(1e-6)))
    out = raw.copy()
    out[..., 4:] = np.concatenate([rot, trans[..., None]], -1).reshape(*raw.shape[:2], 12)
    return out
