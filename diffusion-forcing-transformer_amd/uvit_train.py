"""Training units of the UViT3DPose backbone on the MI355X engine, composed op by op over the C ABI (`dfot_op_*`): every value is
computed by a HIP kernel; Python only sequences the calls (the reference's own structure is Python) and owns the buffers.

``TransformerBlockTrain`` / ``ResBlockTrain``: forward with saved activations and the hand-written backward of one block
(algorithms/dfot/backbones/u_vit/u_vit_blocks.py:57-93,192-281); ``UViT3DPoseTrainer``: the whole backbone (u_vit3d_pose.py:63-131) with the
training step of ``DFoTVideo.training_step`` (continuous v-prediction loss) and a flat-buffer clipped AdamW.
A first, op-by-op driver: correct (parity-tested against torch autograd through the oracle), not yet tuned.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import capi

BF = torch.bfloat16
_S = capi.stream_ptr
_P = capi.ptr
FUSED_PROJ = os.environ.get("DFOT_TRAIN_FUSED_PROJ", "1") != "0"  # A/B: 0 = plain GEMM + separate norm / RoPE and SiLU passes
_PV = capi.ptr_rows  # matrices that may be column blocks of wider ones (the entry point takes the row stride)


def _bf(t: torch.Tensor) -> torch.Tensor:
    out = torch.empty(t.shape, dtype=BF, device="cuda")
    capi.check(capi.lib.dfot_op_f32_to_bf16(_P(t.contiguous()), _P(out), t.numel(), _S()))
    return out


def transpose(src: torch.Tensor, pad_rows_to: int = 1) -> torch.Tensor:
    """[R][C] bf16 -> [C (zero-padded up to a multiple of pad_rows_to)][R]"""
    r, c = src.shape
    cp = -(-c // pad_rows_to) * pad_rows_to
    dst = torch.zeros(cp, r, dtype=BF, device="cuda") if cp != c else torch.empty(c, r, dtype=BF, device="cuda")
    capi.check(capi.lib.dfot_op_transpose_bf16(_P(src), _P(dst), r, c, _S()))
    return dst


def gemm_bf16(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a [M][K] @ w [N][K]^T (+ bias) -> bf16 [M][N]"""
    m, k = a.shape
    out = torch.empty(m, w.shape[0], dtype=BF, device="cuda")
    capi.check(capi.lib.dfot_op_gemm_bf16(_P(a), a.stride(0), _P(w), _P(bias), _P(out), out.stride(0), m, w.shape[0], k, _S()))
    return out


def gemm_bf16_frames(a: torch.Tensor, w: torch.Tensor, frame_bias: torch.Tensor, rows_per_frame: int) -> torch.Tensor:
    """a [M][K] @ w [N][K]^T + frame_bias[row // rows_per_frame] -> bf16 [M][N]   (frame_bias fp32 [M / rows_per_frame][N], contiguous)"""
    m, k = a.shape
    out = torch.empty(m, w.shape[0], dtype=BF, device="cuda")
    capi.check(capi.lib.dfot_op_gemm_bf16_frame_bias(_PV(a), a.stride(0), _P(w), _P(frame_bias), rows_per_frame, _P(out), out.stride(0), m, w.shape[0], k,
                                                     _S()))
    return out


def gemm_f32(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 [M][N] = a w^T (+ bias) (+ resid); out may be resid itself (accumulate in place)"""
    m, k = a.shape
    if out is None:
        out = torch.empty(m, w.shape[0], dtype=torch.float32, device="cuda")
    capi.check(capi.lib.dfot_op_gemm_f32(_PV(a), a.stride(0), _P(w), _P(bias), _P(resid), _P(out), out.stride(0), m, w.shape[0], k, _S()))
    return out


def colsum(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty(x.shape[1], dtype=torch.float32, device="cuda")
    capi.check(capi.lib.dfot_op_colsum_bf16(_PV(x), x.stride(0), _P(out), x.shape[0], x.shape[1], _S()))
    return out


def wgrad(dy: torch.Tensor, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW [M][N] fp32 = dy^T x over the token axis; dy [rows][M], x [rows][N] bf16; out: a contiguous [M][N] fp32 destination"""
    rows, m = dy.shape
    n = x.shape[1]
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device="cuda")
    capi.check(capi.lib.dfot_op_wgrad_nt(_PV(dy), dy.stride(0), _PV(x), x.stride(0), _P(out), m, n, rows, 0, _S()))  # 0: tile form / K slices by shape
    return out


def sgemm(a: torch.Tensor, b: torch.Tensor, ta: bool = False, tb: bool = False, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """fp32 op(a) @ op(b) for weight-sized matrices (op = transpose when ta / tb), no bf16 rounding; `out` (+)= when accumulate"""
    m, k = (a.shape[1], a.shape[0]) if ta else a.shape
    n = b.shape[0] if tb else b.shape[1]
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device="cuda")
    sa = (1, a.stride(0)) if ta else (a.stride(0), 1)     # (stride over i, stride over k)
    sb = (1, b.stride(0)) if tb else (b.stride(0), 1)     # (stride over k, stride over j)
    capi.check(capi.lib.dfot_op_sgemm(_P(a), sa[0], sa[1], _P(b), sb[0], sb[1], _P(out), out.stride(0), m, n, k, 1 if accumulate else 0, _S()))
    return out


def split_bf16(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 x = hi + lo with both parts in bf16 (lo carries the next 8 mantissa bits): the operands of `wprod`"""
    x = x.contiguous()
    hi, lo = torch.empty(x.shape, dtype=BF, device="cuda"), torch.empty(x.shape, dtype=BF, device="cuda")
    capi.check(capi.lib.dfot_op_split_bf16(_P(x), _P(hi), _P(lo), x.numel(), _S()))
    return hi, lo


def wprod(a: Tuple[torch.Tensor, torch.Tensor], b: Tuple[torch.Tensor, torch.Tensor], out: Optional[torch.Tensor] = None,
          accumulate: bool = False) -> torch.Tensor:
    """fp32 [M][N] (+)= A B^T for weight-sized operands given as split_bf16 pairs ([M][K], [N][K]; M a multiple of 128, K of 64):
    Ah Bh + Ah Bl + Al Bh on the bf16 MFMA GEMM with fp32 accumulation -- relative error ~2^-16, against 2^-9 for a plain bf16 product"""
    (ah, al), (bh, bl) = a, b
    out = gemm_f32(ah, bh, resid=out if accumulate else None, out=out)
    gemm_f32(ah, bl, resid=out, out=out)
    return gemm_f32(al, bh, resid=out, out=out)


def wprod_t(a: Tuple[torch.Tensor, torch.Tensor], b: Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
    """fp32 [M][N] = A^T B for split_bf16 pairs A [rows][M], B [rows][N] that share their LONG axis (rows >> M, N; rows a multiple of
    64): the token-axis weight-gradient kernel with K slices instead of a GEMM with a handful of output tiles"""
    (ah, al), (bh, bl) = a, b
    return wgrad(ah, bh) + wgrad(ah, bl) + wgrad(al, bh)


def _pad_rows(x: torch.Tensor, mult: int = 128) -> torch.Tensor:
    r = -(-x.shape[0] // mult) * mult
    if r == x.shape[0]:
        return x.contiguous()
    o = torch.zeros(r, x.shape[1], dtype=x.dtype, device=x.device)
    o[: x.shape[0]] = x
    return o


def frame_sums(src: torch.Tensor, bt: int, pixels: int) -> torch.Tensor:
    """fp32 [bt][n]: per-frame column sums of bf16 src [bt * pixels][n]"""
    n = src.shape[1]
    out = torch.empty(bt, n, dtype=torch.float32, device="cuda")
    capi.check(capi.lib.dfot_op_frame_sums_bf16(_PV(src), src.stride(0), _P(out), bt, pixels, n, _S()))
    return out


def rope_table(head_dim: int, sizes: Tuple[int, int, int], theta: float = 10000.0) -> torch.Tensor:
    """(cos, sin) [T*H*W][head_dim/2][2] of RotaryEmbedding3D (embeddings.py:251-277): per-axis share of the head dim"""
    half = head_dim // 2
    q, r = divmod(half, 3)
    parts = {0: (q, q, q), 1: (q + 1, q, q), 2: (q, q + 1, q + 1)}[r]
    cols = []
    for axis, (p, n) in enumerate(zip(parts, sizes)):
        dim = 2 * p
        inv = 1.0 / (theta ** (torch.arange(0, dim, 2)[:p].float() / dim))
        ang = torch.arange(n, dtype=torch.float32)[:, None] * inv[None, :]
        view = [1, 1, 1, p]
        view[axis] = n
        cols.append(ang.view(*view).expand(*sizes, p))
    ang = torch.cat(cols, dim=-1).reshape(-1, half)
    return torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous().cuda()


_ADAM_MAX: Dict[tuple, float] = {}


def _adam_max_update(b1: float, b2: float) -> float:
    """sup over steps t of |m_hat| / sqrt(v_hat) for ANY gradient sequence (needs b1^2 < b2):  m_t^2 <= (1 - b1)^2 S_t v_t / (1 - b2) with
    S_t = sum_{k<t} (b1^2 / b2)^k, times the bias corrections sqrt(1 - b2^t) / (1 - b1^t).  1.0 at t = 1; 2.35 for (0.9, 0.99)."""
    key = (float(b1), float(b2))
    if key not in _ADAM_MAX:
        r = b1 * b1 / b2
        t = np.arange(1, 1 << 16, dtype=np.float64)
        s_t = (1.0 - r ** t) / (1.0 - r)
        with np.errstate(divide="ignore", invalid="ignore"):
            corr = np.sqrt(1.0 - b2 ** t) / np.where(b1 > 0, 1.0 - b1 ** t, 1.0)
        sup = float(np.max((1.0 - b1) * np.sqrt(s_t / (1.0 - b2)) * corr))
        _ADAM_MAX[key] = max(sup, (1.0 - b1) / math.sqrt((1.0 - b2) * (1.0 - r)))
    return _ADAM_MAX[key]


class TransformerBlockTrain:
    """One UViT TransformerBlock: parameters under the reference's names (fp32 master copies), forward / backward over the C ABI."""

    NAMES = ("norm.emb_layer.weight", "norm.emb_layer.bias", "norm.norm.weight", "fused_attn_mlp_proj.weight", "fused_attn_mlp_proj.bias",
             "q_norm.weight", "k_norm.weight", "attn_out.weight", "attn_out.bias", "mlp_out.2.weight", "mlp_out.2.bias")

    def __init__(self, params: Dict[str, torch.Tensor], prefix: str, channels: int, heads: int, rope: torch.Tensor, eps: float = 1e-6):
        self.c, self.heads, self.d, self.eps, self.rope, self.prefix = channels, heads, channels // heads, eps, rope, prefix
        if self.d not in (64, 128):
            raise ValueError(f"head dim {self.d} not in (64, 128)")
        self.p = {n: params[f"{prefix}.{n}"].detach().to(device="cuda", dtype=torch.float32).contiguous() for n in self.NAMES}
        self.grads: Dict[str, torch.Tensor] = {}
        # bound of |q.k| log2(e) / sqrt(d) from the q_norm / k_norm weights (set by the trainer, see _refresh_score_bounds); inf = take
        # the running-max attention kernel
        self.score_bound = float("inf")
        self._attn_pool: List[Optional[torch.Tensor]] = [None]  # replaced by the trainer's shared one
        self.dM_out: Optional[torch.Tensor] = None               # folded FiLM: where the trainer wants this block's dM (a slice of the level's)
        self.sync()

    def sync(self) -> None:
        """bf16 compute copies: [out][in] for the forward, [in][out] for the data gradients"""
        p = self.p
        self.w_e, self.w_f = _bf(p["norm.emb_layer.weight"]), _bf(p["fused_attn_mlp_proj.weight"])
        self.w_out = _bf(torch.cat([p["attn_out.weight"], p["mlp_out.2.weight"]], dim=1))  # [C][5C]: the two output Linears as one GEMM
        self.b_out = (p["attn_out.bias"] + p["mlp_out.2.bias"]).contiguous()
        self.w_eT, self.w_fT, self.w_outT = transpose(self.w_e), transpose(self.w_f), transpose(self.w_out)

    def forward(self, x: torch.Tensor, emb: Optional[torch.Tensor], batch: int, mlp_mask: Optional[torch.Tensor] = None,
                fold: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, int]] = None) -> torch.Tensor:
        """x fp32 [B*N][C] (residual stream), emb bf16 [B*N][E] (per-token conditioning embedding); returns y fp32.
        mlp_mask: bf16 [B*N][4C] holding 0 or 1 / (1 - p): the nn.Dropout(p) after the MLP branch's SiLU (u_vit_blocks.py:230-234, training).
        fold = (patches bf16 [B*N][K], M bf16 [2C][K], frame table fp32 [frames][2C], tokens per frame) instead of emb: the trainer folded
        norm.emb_layer into the pose patch embedding (film = patches M^T + table[frame], one GEMM; UViT3DPoseTrainer.sync); the block then leaves
        dM = dfilm^T patches and dv = per-frame sums of dfilm (self.dM, self.dv) instead of the emb_layer / embedding gradients"""
        c, hds, d, p = self.c, self.heads, self.d, self.p
        rows = x.shape[0]
        ntok = rows // batch
        lib = capi.lib
        film = gemm_bf16(emb, self.w_e, p["norm.emb_layer.bias"]) if fold is None else gemm_bf16_frames(fold[0], fold[1], fold[2], fold[3])
        xn = torch.empty(rows, c, dtype=BF, device="cuda")
        capi.check(lib.dfot_op_rms_film_fwd(_P(x), _P(p["norm.norm.weight"]), _P(film), self.eps, _P(xn), rows, c, _S()))
        q, k, v = (torch.empty(batch, hds, ntok, d, dtype=BF, device="cuda") for _ in range(3))
        cat = torch.empty(rows, 5 * c, dtype=BF, device="cuda")  # [attention output | SiLU(mlp_h)]
        if FUSED_PROJ and rows % 128 == 0:
            # one launch: the raw projection is kept for the backward while the GEMM epilogue applies QK-norm + RoPE (q, k, v in the attention
            # layout) and SiLU (MLP half, straight into cat) -- no norm / RoPE pass and no SiLU pass over the 7C-wide projection
            fused = torch.empty(rows, 7 * c, dtype=BF, device="cuda")
            capi.check(lib.dfot_op_fused_proj_train(_P(xn), c, _P(self.w_f), _P(p["fused_attn_mlp_proj.bias"]), _P(p["q_norm.weight"]),
                                                    _P(p["k_norm.weight"]), _P(self.rope), self.eps, math.log2(math.e) / math.sqrt(d), _P(fused),
                                                    _P(q), _P(k), _P(v), _P(cat), 5 * c, c, rows, ntok, hds, d, _S()))
        else:
            fused = gemm_bf16(xn, self.w_f, p["fused_attn_mlp_proj.bias"])
            capi.check(lib.dfot_op_qknorm_rope_fwd(_P(fused), 7 * c, _P(p["q_norm.weight"]), _P(p["k_norm.weight"]), _P(self.rope), self.eps,
                                                   math.log2(math.e) / math.sqrt(d), _P(q), _P(k), _P(v), rows, ntok, hds, d, _S()))
            capi.check(lib.dfot_op_silu_cols(_P(fused), 7 * c, 3 * c, None, 0, 0, _P(cat), 5 * c, c, rows, 4 * c, _S()))
        lse = torch.empty(batch, hds, ntok, dtype=torch.float32, device="cuda")
        # key-split partial rows of the attention tail: ONE buffer per trainer (its blocks run one after the other on one stream), grown
        # here through the caching allocator -- never the library's process-wide block, never shared with another trainer
        need = int(lib.dfot_op_attention_scratch_bytes(batch, hds, ntok, d))
        pool = self._attn_pool
        if need and (pool[0] is None or pool[0].numel() < need):
            pool[0] = torch.empty(need, dtype=torch.uint8, device="cuda")
        capi.check(lib.dfot_op_attention_fwd_lse_bounded(_P(q), _P(k), _P(v), _P(cat), 5 * c, _P(lse), batch, hds, ntok, d,
                                                         float(self.score_bound), _P(pool[0]) if need else None, need, _S()))
        if mlp_mask is not None:
            capi.check(lib.dfot_op_mul_cols(_P(cat), 5 * c, c, _P(mlp_mask), rows, 4 * c, _S()))
        y = gemm_f32(cat, self.w_out, self.b_out, resid=x)
        self.saved = dict(x=x, emb=emb, fold=fold, film=film, xn=xn, fused=fused, q=q, k=k, v=v, cat=cat, lse=lse, batch=batch, mlp_mask=mlp_mask)
        return y

    def drop_saved(self) -> None:
        """gradient checkpointing (torch.utils.checkpoint around the block, u_vit3d.py:237-243): keep the block's inputs only"""
        self.saved = {k: self.saved[k] for k in ("x", "emb", "fold", "batch", "mlp_mask")}

    def backward(self, dy: torch.Tensor, demb_acc: Optional[torch.Tensor] = None, dy_bf: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """dy fp32 [B*N][C] -> (dx fp32, demb fp32 [B*N][E]); parameter gradients in self.grads (reference names).
        demb_acc: the level's embedding-gradient accumulator, added to in the GEMM epilogue (no separate pass).
        dy_bf: dy already in bf16 (the block above wrote it next to its dx: self.dx_bf), else it is cast here"""
        ck = None
        if "xn" not in self.saved:  # gradient checkpointing: only the block's inputs were kept -- run its forward again (same kernels, same
            ck = self.saved        # dropout mask: bit-identical activations), then the ordinary backward
            self.forward(ck["x"], ck["emb"], ck["batch"], ck["mlp_mask"], ck["fold"])
        s, c, hds, d, p, lib = self.saved, self.c, self.heads, self.d, self.p, capi.lib
        rows, batch = dy.shape[0], s["batch"]
        ntok = rows // batch
        dyb = dy_bf if dy_bf is not None else _bf(dy)
        dcat = gemm_bf16(dyb, self.w_outT)                                   # [rows][5C]
        dw_out = wgrad(dyb, s["cat"])                                        # [C][5C]
        db_out = colsum(dyb)
        dq, dk, dv = (torch.empty(batch, hds, ntok, d, dtype=BF, device="cuda") for _ in range(3))
        delta = torch.empty_like(s["lse"])
        capi.check(lib.dfot_op_attention_bwd_lse(_P(s["q"]), _P(s["k"]), _P(s["v"]), _P(s["cat"]), _P(dcat), 5 * c, _P(s["lse"]), _P(delta),
                                                 _P(dq), _P(dk), _P(dv), batch, hds, ntok, d, _S()))
        dfused = torch.empty(rows, 7 * c, dtype=BF, device="cuda")
        dqw, dkw = torch.empty(d, device="cuda"), torch.empty(d, device="cuda")
        capi.check(lib.dfot_op_qknorm_rope_bwd(_P(s["fused"]), 7 * c, _P(dq), _P(dk), _P(dv), _P(p["q_norm.weight"]), _P(p["k_norm.weight"]),
                                               _P(self.rope), self.eps, _P(dfused), 7 * c, _P(dqw), _P(dkw), rows, ntok, hds, d, _S()))
        if s["mlp_mask"] is not None:  # d(dropout): the same mask on the gradient of the dropped activations
            capi.check(lib.dfot_op_mul_cols(_P(dcat), 5 * c, c, _P(s["mlp_mask"]), rows, 4 * c, _S()))
        capi.check(lib.dfot_op_silu_cols(_P(s["fused"]), 7 * c, 3 * c, _P(dcat), 5 * c, c, _P(dfused), 7 * c, 3 * c, rows, 4 * c, _S()))
        dxn = gemm_f32(dfused, self.w_fT)                                    # [rows][C]
        dw_f, db_f = wgrad(dfused, s["xn"]), colsum(dfused)
        dx = torch.empty_like(dy)                                            # residual path + the norm's input gradient, one pass
        dfilm = torch.empty(rows, 2 * c, dtype=BF, device="cuda")
        dnw = torch.empty(c, device="cuda")
        self.dx_bf = torch.empty(rows, c, dtype=BF, device="cuda")
        fold = s["fold"]
        capi.check(lib.dfot_op_rms_film_bwd_res(_P(s["x"]), _P(dxn), _P(p["norm.norm.weight"]), _P(s["film"]), self.eps, _P(dy), _P(dx), _P(self.dx_bf),
                                                _P(dfilm), _P(dnw), rows, c, _S()))
        if fold is None:
            demb = gemm_f32(dfilm, self.w_eT, resid=demb_acc, out=demb_acc)      # [rows][E]
            self.grads = {"norm.emb_layer.weight": wgrad(dfilm, s["emb"]), "norm.emb_layer.bias": colsum(dfilm)}
        else:
            demb = None
            self.dM, self.dv = wgrad(dfilm, fold[0], out=self.dM_out), frame_sums(dfilm, rows // fold[3], fold[3])
            self.grads = {}
        self.grads.update({
            "norm.norm.weight": dnw,
            "fused_attn_mlp_proj.weight": dw_f, "fused_attn_mlp_proj.bias": db_f, "q_norm.weight": dqw, "k_norm.weight": dkw,
            "attn_out.weight": dw_out[:, :c].contiguous(), "attn_out.bias": db_out, "mlp_out.2.weight": dw_out[:, c:].contiguous(),
            "mlp_out.2.bias": db_out.clone(),
        })
        if ck is not None:
            self.saved = ck  # release the recomputed activations
        return dx, demb


def conv3x3(a: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, bt: int, h: int, w: int, cin: int, cout: int,
            resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a bf16 [BT*H*W][Cin] channels-last -> fp32 [BT*H*W][Cout] (+ resid)"""
    out = torch.empty(bt * h * w, cout, dtype=torch.float32, device="cuda")
    capi.check(capi.lib.dfot_op_conv3x3_f32(_P(a), _P(w_packed), _P(bias), _P(resid), _P(out), bt, h, w, cin, cout, _S()))
    return out


def conv3x3_backward(x: torch.Tensor, dy: torch.Tensor, w: torch.Tensor, bt: int, h: int, wd: int, cin: int, cout: int, dx_bf16: bool = False):
    """x bf16 [pix][Cin], dy bf16 [pix][Cout], w fp32 [Cout][Cin][3][3] -> (dx [pix][Cin] fp32, or bf16 when it only feeds a GroupNorm
    backward, dW fp32, db fp32)"""
    dx = torch.empty(bt * h * wd, cin, dtype=BF if dx_bf16 else torch.float32, device="cuda")
    dw, db = torch.empty_like(w), torch.empty(cout, dtype=torch.float32, device="cuda")
    capi.check(capi.lib.dfot_op_conv3x3_bwd2(_P(x), _P(dy), _P(w), None if dx_bf16 else _P(dx), _P(dx) if dx_bf16 else None, _P(dw), _P(db), bt, h, wd,
                                             cin, cout, _S()))
    return dx, dw, db


def pack_conv(w: torch.Tensor) -> torch.Tensor:
    co, ci = w.shape[:2]
    out = torch.empty(co, 9 * ci, dtype=BF, device="cuda")
    capi.check(capi.lib.dfot_op_pack_conv3(_P(w), _P(out), co, ci, 0, _S()))
    return out


class ResBlockTrain:
    """One UViT ResBlock (u_vit_blocks.py:57-93) on channels-last streams: GN-SiLU-conv, per-pixel FiLM, GN-FiLM-SiLU-conv, residual."""

    NAMES = ("emb_layer.weight", "emb_layer.bias", "in_layers.0.weight", "in_layers.0.bias", "in_layers.2.weight", "in_layers.2.bias",
             "out_norm.weight", "out_norm.bias", "out_rest.1.weight", "out_rest.1.bias")

    def __init__(self, params: Dict[str, torch.Tensor], prefix: str, channels: int, eps: float = 1e-6):
        self.c, self.eps, self.prefix = channels, eps, prefix
        self.p = {n: params[f"{prefix}.{n}"].detach().to(device="cuda", dtype=torch.float32).contiguous() for n in self.NAMES}
        self.grads: Dict[str, torch.Tensor] = {}
        self.sync()

    def sync(self) -> None:
        p = self.p
        self.w_e = _bf(p["emb_layer.weight"].flatten(1))  # 1x1 conv = Linear over the embedding channels
        self.w_eT = transpose(self.w_e)
        self.w1, self.w2 = pack_conv(p["in_layers.2.weight"]), pack_conv(p["out_rest.1.weight"])

    def forward(self, x: torch.Tensor, emb: Optional[torch.Tensor], bt: int, h: int, w: int, film: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x fp32 [BT*H*W][C]; either emb bf16 [BT*H*W][E] (the block projects it: film = emb_layer(emb)), or the projection itself -- film
        bf16 [BT*H*W][2C], possibly a column block of a level-wide matrix -- when the trainer folded emb_layer into the pose patch embedding
        (UViT3DPoseTrainer.forward); the emb_layer gradients are then the trainer's to compute from the block's dfilm"""
        c, p, lib, P = self.c, self.p, capi.lib, h * w
        st1, st2 = (torch.empty(bt, 32, 2, dtype=torch.float32, device="cuda") for _ in range(2))
        h1 = torch.empty(bt * P, c, dtype=BF, device="cuda")
        capi.check(lib.dfot_op_gn_silu_fwd(_P(x), _P(p["in_layers.0.weight"]), _P(p["in_layers.0.bias"]), None, self.eps, _P(h1), _P(st1), bt, P, c, _S()))
        c1 = conv3x3(h1, self.w1, p["in_layers.2.bias"], bt, h, w, c, c)
        folded = film is not None
        if not folded:
            film = gemm_bf16(emb, self.w_e, p["emb_layer.bias"])
        h2 = torch.empty(bt * P, c, dtype=BF, device="cuda")
        capi.check(lib.dfot_op_gn_silu_fwd2(_P(c1), _P(p["out_norm.weight"]), _P(p["out_norm.bias"]), _PV(film), film.stride(0), self.eps, _P(h2), _P(st2),
                                            bt, P, c, _S()))
        y = conv3x3(h2, self.w2, p["out_rest.1.bias"], bt, h, w, c, c, resid=x)
        self.saved = dict(x=x, emb=emb, h1=h1, c1=c1, film=film, folded=folded, h2=h2, st1=st1, st2=st2, geom=(bt, h, w))
        return y

    def drop_saved(self) -> None:
        keep = ("x", "emb", "geom", "folded") + (("film",) if self.saved["folded"] else ())
        self.saved = {k: self.saved[k] for k in keep}

    def backward(self, dy: torch.Tensor, demb_acc: Optional[torch.Tensor] = None,
                 dfilm_out: Optional[torch.Tensor] = None, dy_bf: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """dfilm_out: a [rows][2C] bf16 column block of the level's FiLM-gradient matrix; the block then leaves the embedding gradient
        (dfilm W_e over the level's concatenated K) to the caller instead of adding its own [rows][E] product to demb_acc.
        dy_bf: dy already in bf16 (the block above left it in self.dx_bf), else it is cast here"""
        ck = None
        if "h1" not in self.saved:  # gradient checkpointing (see TransformerBlockTrain.backward)
            ck = self.saved
            self.forward(ck["x"], ck["emb"], *ck["geom"], film=ck.get("film"))
        s, c, p, lib = self.saved, self.c, self.p, capi.lib
        bt, h, w = s["geom"]
        P = h * w
        # both convolutions' data gradients feed one GroupNorm backward each and nothing else: bf16 (what autocast leaves there in the reference)
        dh2, dw2, db2 = conv3x3_backward(s["h2"], dy_bf if dy_bf is not None else _bf(dy), p["out_rest.1.weight"], bt, h, w, c, c, dx_bf16=True)
        dfilm = dfilm_out if dfilm_out is not None else torch.empty(bt * P, 2 * c, dtype=BF, device="cuda")
        dg2, dbe2, dg1, dbe1 = (torch.empty(c, dtype=torch.float32, device="cuda") for _ in range(4))
        # the gradient of the first convolution's output only feeds that convolution's data / weight gradients: bf16 alone
        dc1 = torch.empty(bt * P, c, dtype=BF, device="cuda")
        capi.check(lib.dfot_op_gn_silu_bwd6(_P(s["c1"]), _P(dh2), _P(s["st2"]), _P(p["out_norm.weight"]), _P(p["out_norm.bias"]), _PV(s["film"]),
                                            s["film"].stride(0), None, None, _P(dc1), _PV(dfilm), dfilm.stride(0), _P(dg2), _P(dbe2), bt, P, c, _S()))
        demb = None if (dfilm_out is not None or s["folded"]) else gemm_f32(dfilm, self.w_eT, resid=demb_acc, out=demb_acc)
        dh1, dw1, db1 = conv3x3_backward(s["h1"], dc1, p["in_layers.2.weight"], bt, h, w, c, c, dx_bf16=True)
        # dx = dy (residual path) + the first norm's input gradient, in fp32 for the stream and in bf16 for the block below
        dx = torch.empty_like(dy)
        self.dx_bf = torch.empty(bt * P, c, dtype=BF, device="cuda")
        capi.check(lib.dfot_op_gn_silu_bwd5(_P(s["x"]), _P(dh1), _P(s["st1"]), _P(p["in_layers.0.weight"]), _P(p["in_layers.0.bias"]), None, _P(dy),
                                            _P(dx), _P(self.dx_bf), None, 0, _P(dg1), _P(dbe1), bt, P, c, _S()))
        self.grads = {} if s["folded"] else {"emb_layer.weight": wgrad(dfilm, s["emb"]).view_as(p["emb_layer.weight"]), "emb_layer.bias": colsum(dfilm)}
        self.grads.update({
            "in_layers.0.weight": dg1, "in_layers.0.bias": dbe1, "in_layers.2.weight": dw1, "in_layers.2.bias": db1,
            "out_norm.weight": dg2, "out_norm.bias": dbe2, "out_rest.1.weight": dw2, "out_rest.1.bias": db2,
        })
        if ck is not None:
            self.saved = ck  # release the recomputed activations
        return dx, demb


def _axpy(a: torch.Tensor, b: torch.Tensor, alpha: float = 1.0) -> None:
    capi.check(capi.lib.dfot_op_axpy(_P(a), _P(b), alpha, a.numel(), _S()))


def _silu(src: torch.Tensor, grad: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 [rows][n]: SiLU(src), or grad * SiLU'(src)"""
    rows, n = src.shape
    out = torch.empty_like(src)
    capi.check(capi.lib.dfot_op_silu_cols(_P(src), n, 0, _P(grad), n, 0, _P(out), n, 0, rows, n, _S()))
    return out


class UViT3DPoseTrainer:
    """Forward with saved activations and hand-written backward of the whole UViT3DPose backbone
    (algorithms/dfot/backbones/u_vit/u_vit3d_pose.py:63-131, u_vit3d.py:30-185), composed op by op over the C ABI.
    `cfg` carries the fields of u_vit3d_pose.yaml (channels, emb_channels, patch_size 2, block_types, num_updown_blocks, num_mid_blocks,
    num_heads, resolution, max_tokens, in_channels 3, cond_dim 180, noise_dim 256); four levels, as the embedding pyramid assumes."""

    def __init__(self, params: Dict[str, torch.Tensor], cfg):
        g = lambda k, d=None: (cfg[k] if isinstance(cfg, dict) else getattr(cfg, k, d)) if (k in cfg if isinstance(cfg, dict) else hasattr(cfg, k)) else d
        self.ch = list(g("channels"))
        self.e, self.ps, self.heads = int(g("emb_channels")), int(g("patch_size", 2)), int(g("num_heads"))
        self.types, self.nud, self.nmid = list(g("block_types")), list(g("num_updown_blocks")), int(g("num_mid_blocks"))
        self.cin, self.res, self.T = int(g("in_channels", 3)), int(g("resolution")), int(g("max_tokens"))
        self.cdim, self.ndim, self.eps, theta = int(g("cond_dim", 180)), int(g("noise_dim", 256)), float(g("eps", 1e-6)), float(g("rope_theta", 10000.0))
        if len(self.ch) != 4 or self.ps != 2:
            raise ValueError("UViT3DPoseTrainer: four levels and patch size 2 (u_vit3d_pose.yaml)")
        self.r = [self.res // self.ps // (2 ** l) for l in range(4)]
        self.names = [n for n in params if not n.endswith(("timesteps.freqs", "timesteps.phases"))]
        # trainable parameters are views into ONE flat fp32 buffer (reference order, 16-byte aligned): the optimizer is one kernel over it
        # and data parallelism one all-reduce of the flat gradient buffer; the Fourier buffers (persistent, not trained) stay apart
        self.layout: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for n in self.names:
            self.layout[n] = (off, tuple(params[n].shape))
            off += -(-params[n].numel() // 4) * 4
        self.numel = off
        self.flat = torch.zeros(off, device="cuda", dtype=torch.float32)
        self.flat_grads = torch.zeros_like(self.flat)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self._sumsq = torch.zeros(1, device="cuda")
        self.step_count = 0
        self.p = {}
        for n, t in params.items():
            if n in self.layout:
                o, shp = self.layout[n]
                v = self.flat[o: o + t.numel()].view(shp)
                v.copy_(t.detach().to(device="cuda", dtype=torch.float32))
                self.p[n] = v
            else:
                self.p[n] = t.detach().to(device="cuda", dtype=torch.float32).contiguous()
        rope = {l: rope_table(self.ch[l] // self.heads, (self.T, self.r[l], self.r[l]), theta) for l in range(4) if self.types[l] == "TransformerBlock"}

        def block(prefix, lvl):
            if self.types[lvl] == "ResBlock":
                return ResBlockTrain(self.p, prefix, self.ch[lvl], self.eps)
            return TransformerBlockTrain(self.p, prefix, self.ch[lvl], self.heads, rope[lvl], self.eps)
        self.down = [[block(f"down_blocks.{l}.{i}", l) for i in range(n)] for l, n in enumerate(self.nud)]
        self.mid = [block(f"mid_blocks.{i}", 3) for i in range(self.nmid)]
        self.up = [[block(f"up_blocks.{j}.{i + 1}", l) for i in range(self.nud[l])] for j, l in enumerate((2, 1, 0))]
        attn_pool: List[Optional[torch.Tensor]] = [None]
        for b in self._blocks():
            if isinstance(b, TransformerBlockTrain):
                b._attn_pool = attn_pool
        self.grads: Dict[str, torch.Tensor] = {}
        # block_dropouts of u_vit3d_pose.yaml ([0, 0, 0.1, 0.1]); applied only when a CUDA generator is set (training with dropout on)
        self.block_dropouts = list(g("block_dropouts", [0.0] * 4))
        self.dropout_generator: Optional[torch.Generator] = None
        # use_checkpointing of u_vit3d.yaml, per level ([false, false, false, true] for RE10K training, realestate10k_video_generation.yaml:44):
        # the blocks of such a level keep only their inputs and are run forward again inside the backward
        self.use_checkpointing = [bool(v) for v in g("use_checkpointing", [False] * 4)]
        # optional trainer state (experiments/simple_video_generation.py): EMA shadow weights, gradient accumulation
        self.ema: Optional[torch.Tensor] = None
        self.ema_decay = 0.0
        self._acc: Optional[torch.Tensor] = None
        self._acc_n = 0
        self.sync()

    def sync(self, own_step: Optional[Dict] = None) -> None:
        """re-pack the bf16 operand copies from the fp32 master weights.  own_step: set only by this trainer's own `optimizer_step`
        (its lr / betas / weight_decay bound how far the weights moved); None = the weights were changed from outside"""
        p, e = self.p, self.e
        ne = "noise_level_pos_embedding.embedding."
        self.w1, self.w2 = _bf(p[ne + "linear_1.weight"]), _bf(p[ne + "linear_2.weight"])
        self.w2T = transpose(self.w2)
        self.kpad = -(-self.cdim * 4 // 64) * 64
        wp = torch.zeros(e, self.kpad, device="cuda")
        wp[:, : self.cdim * 4] = p["external_cond_embedding.patch_embedder.proj.weight"].flatten(1)
        self.wp, self.wp32 = _bf(wp), wp
        wo = torch.zeros(self.ch[0], 64, device="cuda")
        wo[:, : self.cin * 4] = p["project_output.proj.weight"].flatten(1)
        self.wo = _bf(wo)                    # [C0][64]: data gradient of the ConvTranspose as a GEMM with K = 64
        self.wd = [pack_conv(p[f"down_blocks.{l}.{n}.conv.weight"]) for l, n in enumerate(self.nud)]
        self.wu = [pack_conv(p[f"up_blocks.{j}.0.conv.weight"]) for j in range(3)]
        for b in self._blocks():
            b.sync()
        self._refresh_score_bounds(own_step)
        # FiLM folded into the pose patch embedding, every level.  emb = PatchEmbed(patches) keep + noise embedding is linear in the
        # patches (and average pools commute with it), and each block's emb_layer is linear in emb, so
        #     film = (W_e W_p) patches_l + W_e (b_p keep + nemb[frame]) + b_e :
        # the per-row GEMM runs over the 768-wide pose patches instead of the 1024-wide embedding, the per-frame part is a [BT][2C] vector,
        # and the backward never forms a per-row embedding gradient (4.3 GB in fp32 at level 0 of config 5; one [rows][E] GEMM per
        # transformer block).  Per level: the blocks in order, their row offsets in the concatenated matrices (= column offsets in a
        # ResBlock level's FiLM-gradient matrix), W_e concatenated [R][E] as a split-bf16 pair, b_e [R], M = W_e W_p bf16
        self.res_cols: Dict[int, int] = {}
        self.fold_blocks: Dict[int, list] = {}
        self.fold_w: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.fold_b32: Dict[int, torch.Tensor] = {}
        self.fold_m: Dict[int, torch.Tensor] = {}
        self.p_split = split_bf16(self.wp32)                                       # W_p [E][kpad]
        self.pT_split = tuple(t.t().contiguous() for t in self.p_split)            # W_p^T [kpad][E]
        for l in range(4):
            blocks = self.mid if l == 3 else self.down[l] + self.up[2 - l]
            if not blocks:
                continue
            key = lambda b: "emb_layer" if isinstance(b, ResBlockTrain) else "norm.emb_layer"
            for i, b in enumerate(blocks):
                self.res_cols[id(b)] = i * 2 * self.ch[l]
            self.fold_blocks[l] = blocks
            w32 = torch.cat([b.p[key(b) + ".weight"].flatten(1) for b in blocks], dim=0).contiguous()
            self.fold_b32[l] = torch.cat([b.p[key(b) + ".bias"] for b in blocks], dim=0).contiguous()
            self.fold_w[l] = split_bf16(w32)
            self.fold_m[l] = _bf(wprod(self.fold_w[l], self.pT_split))

    def _blocks(self):
        return [b for lv in self.down for b in lv] + self.mid + [b for lv in self.up for b in lv]

    def _refresh_score_bounds(self, own_step: Optional[Dict] = None) -> None:
        """The d = 64 blocks may run their forward attention without a running max while  B = sqrt(d) log2(e) * max over rotary pairs of
        max|w_q| * max|w_k|  stays below 64 (csrc/uvit.hip, u_vit_blocks.py:255-262).  What a block is told is an UPPER bound of B that
        holds by construction:
          * any weight change from outside (construction, a checkpoint or state dict copied into a live trainer, the autograd drop-in
            after an external optimizer step, an EMA swap) reaches `sync()` with own_step = None: B and max|w| are computed on the
            device and read back SYNCHRONOUSLY before the next forward can be issued;
          * the trainer's own AdamW step moves every weight by at most  lr * (A + wd * |w|)  with  A = sup_t (1 - b1) sqrt(S_t / (1 - b2))
            sqrt(1 - b2^t) / (1 - b1^t),  S_t = sum_{k<t} (b1^2 / b2)^k  (Cauchy-Schwarz on the two moment sums, bias corrections
            included: `_adam_max_update`; gradient clipping scales g and cancels; eps only shrinks the update), so the host adds that
            drift d to every |w|:  (|w_q| + d)(|w_k| + d) <= |w_q||w_k| + 2 max|w| d + d^2  -- no device read, no
            assumption about how far the weights move.  The exact value is re-read every 256 own steps (one host sync), or at once
            when the drifted bound would cross 64, or when b1^2 >= b2 (no finite A)."""
        blocks = [b for b in self._blocks() if isinstance(b, TransformerBlockTrain) and b.d == 64]
        if not blocks:
            return
        d = blocks[0].d
        kd = math.sqrt(d) * math.log2(math.e)
        exact = own_step is None or not hasattr(self, "_bound_exact")
        if not exact:
            b1, b2 = own_step["betas"]
            if not (0.0 <= b1 * b1 < b2 < 1.0):
                exact = True
            else:
                a = _adam_max_update(b1, b2)
                wmax = max(self._bound_wmax) + self._bound_drift
                self._bound_drift += float(own_step["lr"]) * (a + float(own_step["weight_decay"]) * wmax)
                self._bound_own_steps += 1
                worst = max(e + kd * (2.0 * w * self._bound_drift + self._bound_drift ** 2) for e, w in zip(self._bound_exact, self._bound_wmax))
                was_fast = max(self._bound_exact) < 64.0
                if self._bound_own_steps >= 256 or (was_fast and worst >= 64.0):
                    exact = True
        if exact:
            wq = torch.stack([b.p["q_norm.weight"] for b in blocks]).abs()
            wk = torch.stack([b.p["k_norm.weight"] for b in blocks]).abs()
            pq, pk = wq.view(len(blocks), d // 2, 2).amax(-1), wk.view(len(blocks), d // 2, 2).amax(-1)
            bound = torch.nan_to_num((pq * pk).amax(-1) * kd, nan=float("inf"))
            wmx = torch.nan_to_num(torch.maximum(wq.amax(-1), wk.amax(-1)), nan=float("inf"))
            host = torch.stack([bound, wmx]).tolist()  # synchronous: the forward that follows sees THESE weights' bound
            self._bound_exact, self._bound_wmax = host[0], host[1]
            self._bound_drift, self._bound_own_steps = 0.0, 0
            self._bound_blocks = blocks
            self.bound_exact_reads = getattr(self, "bound_exact_reads", 0) + 1
        dr = self._bound_drift
        for b, e, w in zip(self._bound_blocks, self._bound_exact, self._bound_wmax):
            b.score_bound = (e + kd * (2.0 * w * dr + dr * dr)) * (1.0 + 1e-5)  # the product above was rounded in fp32 on the device

    def _run(self, blocks, x, lvl):
        p = self.block_dropouts[lvl] if self.dropout_generator is not None else 0.0
        for b in blocks:
            c0, c2 = self.res_cols[id(b)], 2 * self.ch[lvl]
            if isinstance(b, ResBlockTrain):
                # the level's ResBlocks share ONE projection GEMM (N = blocks * 2C): the patches are read once, not once per block
                # (level 0 of config 5: 1.6 GB of patches against 0.5 GB of output per block -- the per-block GEMM was HBM-bound)
                if lvl not in self.film_cat:
                    self.film_cat[lvl] = gemm_bf16_frames(self.xl[lvl], self.fold_m[lvl], self.film_vec[lvl], self.r[lvl] * self.r[lvl])
                x = b.forward(x, None, self.bt, self.r[lvl], self.r[lvl], film=self.film_cat[lvl][:, c0: c0 + c2])
            else:
                mask = None
                if p > 0:  # nn.Dropout(p) of the MLP branch: keep with probability 1 - p, scale by 1 / (1 - p)
                    keep = torch.rand(x.shape[0], 4 * self.ch[lvl], device="cuda", generator=self.dropout_generator) >= p
                    mask = (keep.to(torch.float32) / (1.0 - p)).to(BF)
                x = b.forward(x, None, self.B, mask, fold=(self.xl[lvl], self.fold_m[lvl][c0: c0 + c2], self.film_vec[lvl][:, c0: c0 + c2].contiguous(),
                                                          self.r[lvl] * self.r[lvl]))
            if self.use_checkpointing[lvl]:
                b.drop_saved()
        return x

    def forward(self, x: torch.Tensor, noise_levels: torch.Tensor, cond: torch.Tensor, cond_drop: Optional[torch.Tensor] = None) -> torch.Tensor:
        """cond_drop: bool (B,), True = this video's pose embedding is zeroed -- RandomDropoutPatchEmbed's per-video dropout in training
        (external_cond_dropout, embeddings.py:390-428); the caller draws it (torch.rand(B) < p)"""
        lib, p, e, r, ch = capi.lib, self.p, self.e, self.r, self.ch
        self.drop = None if cond_drop is None else cond_drop.to(device="cuda", dtype=torch.uint8).contiguous()
        self.B, t = x.shape[:2]
        bt = self.bt = self.B * t
        if t != self.T:
            raise AssertionError(f"temporal length must be {self.T}, got {t}")
        if cond is None:
            raise AssertionError("camera-pose conditioning is required")
        xd = self.x_in = x.to(device="cuda", dtype=torch.float32).reshape(bt, self.cin, self.res, self.res).contiguous()
        # noise-level embedding (bt rows, padded to the GEMM's 128): Fourier features -> Linear -> SiLU -> Linear
        pre = "noise_level_pos_embedding."
        k = noise_levels.to(device="cuda", dtype=torch.float32).reshape(bt, 1)
        feats = torch.zeros(128 * -(-bt // 128), self.ndim, device="cuda")
        feats[:bt] = (k * p[pre + "timesteps.freqs"] + p[pre + "timesteps.phases"]).cos() * math.sqrt(2.0)
        self.feats = _bf(feats)
        self.l1 = gemm_bf16(self.feats, self.w1, p[pre + "embedding.linear_1.bias"])
        self.a1 = _silu(self.l1)
        nemb = gemm_f32(self.a1, self.w2, p[pre + "embedding.linear_2.bias"])[:bt].contiguous()
        # pose patch embedding + embedding pyramid
        P0 = r[0] * r[0]
        self.patches = torch.zeros(bt * P0, self.kpad, dtype=BF, device="cuda")
        cd = cond.to(device="cuda", dtype=torch.float32).reshape(bt, self.cdim, self.res, self.res).contiguous()
        capi.check(lib.dfot_op_cond_repack(_P(cd), _P(self.patches), bt, self.res, self.cdim, self.kpad, _S()))
        bp = p["external_cond_embedding.patch_embedder.proj.bias"]
        keep = torch.ones(bt, device="cuda") if self.drop is None else (1.0 - self.drop.to(torch.float32)).repeat_interleave(t)
        if self.drop is not None:  # a dropped video's pose embedding is zero: no patches, no patch-embedding bias
            self.patches.view(self.B, -1)[self.drop.bool()] = 0
        self.keep = keep
        # the pyramid is taken over the pose PATCHES (average pools commute with the linear patch embedding): xl[l] [BT * r_l^2][kpad]
        self.xl = [self.patches] + [torch.empty(bt * r[l] * r[l], self.kpad, dtype=BF, device="cuda") for l in (1, 2, 3)]
        capi.check(lib.dfot_op_emb_pyramid(_P(self.xl[0]), _P(self.xl[1]), _P(self.xl[2]), _P(self.xl[3]), bt, r[0], self.kpad, _S()))
        # per-frame part of the embedding, c = b_p keep + nemb, and every block's per-frame FiLM vector W_e c + b_e (a level at a time)
        self.cvec = (nemb + keep[:, None] * bp[None, :]).contiguous()
        self.film_cat: Dict[int, torch.Tensor] = {}
        c_split = split_bf16(_pad_rows(self.cvec))  # the long axis (R) goes to the GEMM's rows: W_e c^T, transposed back
        self.film_vec = {l: wprod(self.fold_w[l], c_split)[:, :bt].t().contiguous() + self.fold_b32[l][None, :] for l in self.fold_blocks}
        # input embedding and the U
        h = torch.empty(bt * P0, ch[0], dtype=torch.float32, device="cuda")
        capi.check(lib.dfot_op_embed_input(_P(xd), _P(p["embed_input.proj.weight"]), _P(p["embed_input.proj.bias"]), _P(h), bt, self.res, self.cin, ch[0], _S()))
        self.before, self.after, self.pooled, self.hsub = [], [], [], [None] * 3
        for l, n in enumerate(self.nud):
            h = self._run(self.down[l], h, l)
            self.before.append(h)
            pooled = torch.empty(bt * r[l + 1] * r[l + 1], ch[l], dtype=BF, device="cuda")
            capi.check(lib.dfot_op_pool2_bf16(_P(h), _P(pooled), bt, r[l], r[l], ch[l], _S()))
            self.pooled.append(pooled)
            h = conv3x3(pooled, self.wd[l], p[f"down_blocks.{l}.{n}.conv.bias"], bt, r[l + 1], r[l + 1], ch[l], ch[l + 1])
            self.after.append(h)
        h = self._run(self.mid, h, 3)
        for j, l in enumerate((2, 1, 0)):
            hs = torch.empty(bt * r[l + 1] * r[l + 1], ch[l + 1], dtype=BF, device="cuda")
            capi.check(lib.dfot_op_sub_bf16(_P(h), _P(self.after[l]), _P(hs), hs.numel(), _S()))
            self.hsub[j] = hs
            tmp = conv3x3(hs, self.wu[j], p[f"up_blocks.{j}.0.conv.bias"], bt, r[l + 1], r[l + 1], ch[l + 1], ch[l])
            h = torch.empty(bt * r[l] * r[l], ch[l], dtype=torch.float32, device="cuda")
            capi.check(lib.dfot_op_upsample_add(_P(tmp), _P(self.before[l]), _P(h), bt, r[l + 1], r[l + 1], ch[l], _S()))  # (h, w) = the coarse map's size
            h = self._run(self.up[j], h, l)
        self.h_final = h
        out = torch.empty(bt, self.cin, self.res, self.res, dtype=torch.float32, device="cuda")
        capi.check(lib.dfot_op_project_output(_P(h), _P(p["project_output.proj.weight"]), _P(p["project_output.proj.bias"]), _P(out), bt, self.res, ch[0],
                                              self.cin, _S()))
        return out.view(self.B, t, self.cin, self.res, self.res)

    def backward(self, d_out: torch.Tensor, reducer=None, input_grad: bool = False) -> Dict[str, torch.Tensor]:
        """reducer (parallel.OverlappedGradReducer): gradients are handed over level by level as they are produced, so their
        all-reduce overlaps the rest of the backward; without it they are only returned.  input_grad: also leave d loss / d x in
        ``self.dx_in`` (B, T, C, H, W; fp32) -- what reconstruction guidance differentiates (discrete_diffusion.py:485-513)"""
        lib, p, e, r, ch, bt = capi.lib, self.p, self.e, self.r, self.ch, self.bt
        G: Dict[str, torch.Tensor] = {}
        handed = set()

        def hand_over():
            if reducer is None:
                return
            for n, gv in G.items():
                if n not in handed:
                    handed.add(n)
                    o, shp = self.layout[n]
                    reducer.add(self.flat_grads[o: o + gv.numel()], gv)
        # FiLM gradients (folded FiLM, see sync): the ResBlocks of a level write theirs side by side into one [rows][blocks * 2C] bf16 matrix,
        # a transformer block leaves dM = dfilm^T patches and the per-frame sums of dfilm; finish_level turns either into weight-sized
        # gradients as soon as the level's last block is done
        dfilm_cat = {l: torch.empty(bt * r[l] * r[l], self.fold_b32[l].numel(), dtype=BF, device="cuda") for l, blocks in self.fold_blocks.items()
                     if isinstance(blocks[0], ResBlockTrain)}
        dP = torch.zeros(e, self.kpad, device="cuda")
        dc = torch.zeros(bt, e, device="cuda")
        kb = 64 * -(-bt // 64)
        # a transformer level's dM [R][kpad]: every block's weight-gradient kernel writes its rows in place
        dM_level = {l: torch.empty(self.fold_b32[l].numel(), self.kpad, device="cuda") for l, blocks in self.fold_blocks.items()
                    if not isinstance(blocks[0], ResBlockTrain)}
        for l, buf in dM_level.items():
            for blk in self.fold_blocks[l]:
                c0 = self.res_cols[id(blk)]
                blk.dM_out = buf[c0: c0 + 2 * ch[l]]
        cT = torch.zeros(e, kb, device="cuda")
        cT[:, :bt] = self.cvec.t()
        cT_split = split_bf16(cT)

        def finish_level(l):
            """the level's emb_layer gradients and its share of dP / dc from dM = dfilm^T patches_l [R][kpad] and dv = per-frame sums of
            dfilm [BT][R]:  dW_e = dM W_p^T + dv^T c,  db_e = sum_frames dv,  dP += W_e^T dM,  dc += dv W_e"""
            blocks = self.fold_blocks[l]
            if l in dfilm_cat:
                dM, dv = wgrad(dfilm_cat[l], self.xl[l]), frame_sums(dfilm_cat[l], bt, r[l] * r[l])
                del dfilm_cat[l]
            else:
                dM, dv = dM_level[l], torch.cat([b.dv for b in blocks], dim=1)
                for b in blocks:
                    b.dM = b.dv = b.dM_out = None
            R = dM.shape[0]
            dM_split = split_bf16(dM)
            dW = wprod(dM_split, self.p_split)
            dvT = torch.zeros(R, kb, device="cuda")
            dvT[:, :bt] = dv.t()
            dvT_split = split_bf16(dvT)
            wprod(dvT_split, cT_split, out=dW, accumulate=True)
            db = sgemm(torch.ones(1, bt, device="cuda"), dv).view(-1)
            dP.add_(wprod_t(self.fold_w[l], dM_split))                      # W_e^T dM: the long axis R is shared -> token-axis kernel
            dc.add_(wprod_t(dvT_split, self.fold_w[l])[:bt])                # dv W_e
            for blk in blocks:
                c0, c2 = self.res_cols[id(blk)], 2 * ch[l]
                res = isinstance(blk, ResBlockTrain)
                name = f"{blk.prefix}.emb_layer" if res else f"{blk.prefix}.norm.emb_layer"
                G[name + ".weight"] = dW[c0: c0 + c2].reshape((c2, e, 1, 1) if res else (c2, e)).clone()  # own storage (outputs of the autograd op must not alias)
                G[name + ".bias"] = db[c0: c0 + c2].clone()

        def run_back(blocks, prefix_fn, dh, lvl):
            dh_bf = None
            for i in reversed(range(len(blocks))):
                # every block also leaves its input gradient in bf16 (dx_bf) for the block below
                if lvl in dfilm_cat:
                    c0 = self.res_cols[id(blocks[i])]
                    dh, _ = blocks[i].backward(dh, None, dfilm_cat[lvl][:, c0: c0 + 2 * ch[lvl]], dh_bf)
                else:
                    dh, _ = blocks[i].backward(dh, None, dh_bf)      # leaves dM / dv for finish_level
                dh_bf = blocks[i].dx_bf
                blocks[i].dx_bf = None
                for n, gv in blocks[i].grads.items():
                    G[f"{prefix_fn(i)}.{n}"] = gv
            return dh
        # output projection (ConvTranspose k = s = 2): per input pixel a Linear C0 -> (co, py, px)
        dout = d_out.to(device="cuda", dtype=torch.float32).reshape(bt, self.cin, self.res, self.res).contiguous()
        P0 = r[0] * r[0]
        dpatch = torch.empty(bt * P0, 64, dtype=BF, device="cuda")
        capi.check(lib.dfot_op_outgrad_gather(_P(dout), _P(dpatch), bt, self.res, self.cin, self.ps, _S()))
        n_out = self.cin * self.ps * self.ps
        G["project_output.proj.weight"] = wgrad(_bf(self.h_final), dpatch)[:, :n_out].reshape(ch[0], self.cin, self.ps, self.ps).contiguous()
        G["project_output.proj.bias"] = colsum(dpatch)[:n_out].view(self.cin, -1).sum(1)
        dh = gemm_f32(dpatch, self.wo)
        dbefore, dsub = [None] * 3, [None] * 3
        for j, l in reversed(list(enumerate((2, 1, 0)))):
            dh = run_back(self.up[j], lambda i, j=j: f"up_blocks.{j}.{i + 1}", dh, l)
            dbefore[l] = dh                                           # skip connection: + before[l]
            dt = torch.empty(bt * r[l + 1] * r[l + 1], ch[l], dtype=torch.float32, device="cuda")
            capi.check(lib.dfot_op_upsample_bwd(_P(dh), _P(dt), bt, r[l], r[l], ch[l], _S()))
            dh, G[f"up_blocks.{j}.0.conv.weight"], G[f"up_blocks.{j}.0.conv.bias"] = conv3x3_backward(
                self.hsub[j], _bf(dt), p[f"up_blocks.{j}.0.conv.weight"], bt, r[l + 1], r[l + 1], ch[l + 1], ch[l])
            dsub[l] = dh                                              # d(h - after[l]): -> h, and minus -> after[l]
            hand_over()
        dh = run_back(self.mid, lambda i: f"mid_blocks.{i}", dh, 3)
        if 3 in self.fold_blocks:
            finish_level(3)
        hand_over()
        for l in (2, 1, 0):
            n = self.nud[l]
            _axpy(dh, dsub[l], -1.0)                                  # after[l] also fed the subtraction on the way up
            dpool, G[f"down_blocks.{l}.{n}.conv.weight"], G[f"down_blocks.{l}.{n}.conv.bias"] = conv3x3_backward(
                self.pooled[l], _bf(dh), p[f"down_blocks.{l}.{n}.conv.weight"], bt, r[l + 1], r[l + 1], ch[l], ch[l + 1])
            dh = dbefore[l].clone()
            capi.check(lib.dfot_op_pool2_bwd(_P(dpool), _P(dh), bt, r[l], r[l], ch[l], _S()))
            dh = run_back(self.down[l], lambda i, l=l: f"down_blocks.{l}.{i}", dh, l)
            if l in self.fold_blocks:
                finish_level(l)
            hand_over()
        dw, db = torch.empty_like(p["embed_input.proj.weight"]), torch.empty(ch[0], device="cuda")
        capi.check(lib.dfot_op_embed_input_wgrad(_P(dh), _P(self.x_in), _P(dw), _P(db), bt, self.res, self.cin, ch[0], self.ps, _S()))
        G["embed_input.proj.weight"], G["embed_input.proj.bias"] = dw, db
        self.dx_in = None
        if input_grad:
            dx = torch.empty_like(self.x_in)
            capi.check(lib.dfot_op_embed_input_dgrad(_P(dh), _P(p["embed_input.proj.weight"]), _P(dx), bt, self.res, self.cin, ch[0], self.ps, _S()))
            self.dx_in = dx.view(self.B, bt // self.B, self.cin, self.res, self.res)
        # pose patch embedding and the per-frame vector c = b_p keep + nemb (dc = the noise embedding's gradient): finish_level left dP, dc
        pe = "external_cond_embedding.patch_embedder.proj."
        G[pe + "weight"] = dP[:, : self.cdim * 4].reshape(e, self.cdim, self.ps, self.ps).contiguous()
        G[pe + "bias"] = sgemm(self.keep.view(1, bt), dc).view(-1)
        dn = torch.zeros(self.feats.shape[0], e, device="cuda")  # rows beyond bt pad the noise-level MLP's GEMMs: they stay zero
        dn[:bt] = dc
        dnb = _bf(dn)
        ne = "noise_level_pos_embedding.embedding."
        G[ne + "linear_2.weight"], G[ne + "linear_2.bias"] = wgrad(dnb, self.a1), colsum(dnb)
        dl1 = _silu(self.l1, gemm_bf16(dnb, self.w2T))
        G[ne + "linear_1.weight"], G[ne + "linear_1.bias"] = wgrad(dl1, self.feats), colsum(dl1)
        hand_over()
        self.grads = G
        return G

    # ------------------------------------------------------------------ training step (ContinuousDiffusion.forward + AdamW)
    def loss_and_grads(self, xs: torch.Tensor, cond: torch.Tensor, t: torch.Tensor, noise: torch.Tensor, masks: Optional[torch.Tensor] = None,
                       diffusion=None, cond_drop: Optional[torch.Tensor] = None, reducer=None) -> torch.Tensor:
        """DFoTVideo.training_step for the pose model (dfot_video.py:41-75, continuous_diffusion.py:140-167): per-token levels t in [0,1],
        x_t = alpha x + sigma eps, v = model(x_t, precond * logsnr, cond), sigmoid-weighted eps-space error averaged with the loss masks;
        then the backward.  cond: processed ray encoding (B,T,180,H,W).  `diffusion`: the DiffusionConfig whose training schedule
        (logsnr_min/max, training_schedule_shift), loss weighting (loss_sigmoid_bias), precond_scale and clip_noise apply (default: the
        reference's RE10K values).  Returns the loss (device scalar)."""
        from .diffusion import DiffusionConfig
        dcfg = diffusion if diffusion is not None else DiffusionConfig()  # schedule limits / shift / loss weighting / preconditioning from the config
        precond_scale, clip_noise = float(dcfg.precond_scale), float(dcfg.clip_noise)
        b, tk = xs.shape[:2]
        f = int(xs[0, 0].numel())
        logsnr, alpha, sigma, weight = dcfg.training_logsnr_tables(t)
        mk = torch.ones(b, tk) if masks is None else masks.detach().float().cpu().view(b, tk)
        tab = torch.stack([alpha, sigma, weight, precond_scale * logsnr, 2.0 * weight * mk / (f * b * tk)]).float().cuda().contiguous()
        x = xs.to(device="cuda", dtype=torch.float32).contiguous()
        eps = noise.to(device="cuda", dtype=torch.float32).clamp(-clip_noise, clip_noise).contiguous()
        x_t = torch.empty_like(x)
        lib = capi.lib
        capi.check(lib.dfot_hg_prepare(_P(x), _P(eps), _P(tab[0]), _P(tab[1]), _P(x_t), b, 1, tk, f, _S()))
        v = self.forward(x_t, tab[3], cond, cond_drop).contiguous()
        per_token = torch.empty(b, tk, device="cuda")
        scratch = torch.empty(int(lib.dfot_vpred_loss_scratch_floats(b, tk, f)), device="cuda")
        capi.check(lib.dfot_vpred_loss(_P(x), _P(eps), _P(v), _P(tab[0]), _P(tab[1]), _P(tab[2]), None, _P(scratch), _P(per_token), b, tk, f, _S()))
        dv = torch.empty_like(x)
        capi.check(lib.dfot_vloss_grad(_P(x), _P(eps), _P(v), _P(tab[0]), _P(tab[1]), _P(tab[4]), _P(dv), b, tk, f, 0, _S()))
        grads = self.backward(dv, reducer)
        if reducer is not None:  # data parallel with the exchange overlapped: the flat buffer receives the MEANS over the ranks
            reducer.finish()
            self._grads_reduced = True
        else:
            # one multi-tensor copy instead of ~430 separate ones (18 us each: 4 ms of launch tails per step)
            dst = [self.flat_grads[o: o + grads[n].numel()] for n, (o, shp) in self.layout.items()]
            src = [grads[n].reshape(-1) for n in self.layout]
            torch._foreach_copy_(dst, src)
            self._grads_reduced = False
        return (per_token * mk.cuda()).mean()

    def accumulate(self) -> None:
        """accumulate_grad_batches (accelerator.accumulate, simple_video_generation.py:260): add the gradients of the last
        loss_and_grads to the running sum; the next optimizer_step uses the MEAN over the accumulated micro-batches"""
        if self._acc is None:
            self._acc = torch.zeros_like(self.flat_grads)
        if self._acc_n == 0:
            self._acc_reduced = True
        # the reference reduces once per optimizer step (accelerator.accumulate runs the micro-batches under no_sync); micro-batches whose
        # gradients were already averaged over the ranks (loss_and_grads with a reducer) need no second exchange -- but only if ALL were
        self._acc_reduced = self._acc_reduced and bool(getattr(self, "_grads_reduced", False))
        self._acc.add_(self.flat_grads)
        self._acc_n += 1

    def optimizer_step(self, lr: float = 5e-5, betas=(0.9, 0.99), eps: float = 1e-8, weight_decay: float = 0.01, max_grad_norm: Optional[float] = 1.0,
                       world_size: int = 1) -> None:
        from . import parallel
        if self._acc_n:
            self.flat_grads.copy_(self._acc).mul_(1.0 / self._acc_n)
            self._acc.zero_()
            self._acc_n = 0
            self._grads_reduced = self._acc_reduced  # local micro-batch gradients: ONE exchange of the accumulated mean, here
        if world_size > 1 and not getattr(self, "_grads_reduced", False):
            parallel.allreduce_mean_(self.flat_grads)
        self.step_count += 1
        self._opt = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        lib = capi.lib
        sumsq = None
        if max_grad_norm is not None:
            capi.check(lib.dfot_sumsq(_P(self.flat_grads), self.numel, _P(self._sumsq), _S()))
            sumsq = self._sumsq
        # EMA shadow weights (algorithms/common/ema.py:21-33: shadow = decay * shadow + (1 - decay) * param after every optimizer step) are
        # updated by the same kernel pass that writes the new parameters
        capi.check(lib.dfot_adamw_step(_P(self.flat), _P(self.flat_grads), _P(self.exp_avg), _P(self.exp_avg_sq), self.numel, lr, betas[0], betas[1], eps,
                                       weight_decay, self.step_count, _P(sumsq), float(max_grad_norm or 0.0), _P(self.ema), float(self.ema_decay), _S()))
        self.sync(own_step=dict(lr=lr, betas=tuple(betas), weight_decay=weight_decay))

    # ------------------------------------------------------------------ EMA and optimizer state (checkpoint / resume), as trainer.DiT3DTrainer
    def enable_ema(self, decay: float) -> None:
        """experiment.ema (algorithms/common/ema.py): shadow weights start as a copy of the parameters"""
        self.ema, self.ema_decay = self.flat.clone(), float(decay)

    def _view(self, name: str, flat: torch.Tensor) -> torch.Tensor:
        o, shp = self.layout[name]
        n = 1
        for d in shp:
            n *= d
        return flat[o: o + n].view(shp)

    def ema_state_dict(self) -> Dict[str, torch.Tensor]:
        """what the reference writes to ema.safetensors (simple_video_generation.py:653-657): the shadow of every trainable parameter"""
        if self.ema is None:
            raise RuntimeError("EMA is not enabled")
        return {k: self._view(k, self.ema).detach().clone() for k in self.layout}

    def load_ema_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        if self.ema is None:
            raise RuntimeError("EMA is not enabled")
        if set(sd.keys()) != set(self.layout.keys()):
            raise ValueError("The provided state_dict does not match the structure of the EMA model.")
        for k, t in sd.items():
            self._view(k, self.ema).copy_(t.to(device="cuda", dtype=torch.float32))

    def optimizer_state_dict(self) -> Dict:
        """torch.optim.AdamW.state_dict() layout (parameter index = position in the reference's parameter order)"""
        opt = getattr(self, "_opt", dict(lr=5e-5, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01))
        state = {i: {"step": torch.tensor(float(self.step_count)), "exp_avg": self._view(k, self.exp_avg).clone(),
                     "exp_avg_sq": self._view(k, self.exp_avg_sq).clone()} for i, k in enumerate(self.layout)} if self.step_count else {}
        group = dict(lr=opt["lr"], betas=opt["betas"], eps=opt["eps"], weight_decay=opt["weight_decay"], amsgrad=False, params=list(range(len(self.layout))))
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: Dict) -> None:
        names = list(self.layout)
        steps = set()
        for i, st in sd.get("state", {}).items():
            k = names[int(i)]
            self._view(k, self.exp_avg).copy_(st["exp_avg"].to("cuda"))
            self._view(k, self.exp_avg_sq).copy_(st["exp_avg_sq"].to("cuda"))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ: the flat optimizer keeps one")
        self.step_count = steps.pop() if steps else 0
        if sd.get("param_groups"):
            g0 = sd["param_groups"][0]
            self._opt = dict(lr=g0["lr"], betas=tuple(g0["betas"]), eps=g0["eps"], weight_decay=g0["weight_decay"])

    def grad_norm(self) -> float:
        capi.check(capi.lib.dfot_sumsq(_P(self.flat_grads), self.numel, _P(self._sumsq), _S()))
        return float(self._sumsq.sqrt().item())

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {n: t.detach().clone() for n, t in self.p.items()}
