"""VideoVAE decoder on the MI355X engine: latents -> frames for the latent datasets (Kinetics-600).

Mirrors the reference's decode path for this model family:
  * ``BaseVideoAlgo._decode`` / ``_run_vae``   algorithms/common/base_pytorch_video_algo.py:553-629   (chunking by
    ``vae.batch_size``, ``decode(y, n_frames) * 0.5 + 0.5``, the ``b t c h w`` layout contract)
  * ``VideoVAE.decode`` / ``_decode``          algorithms/vae/video_vae/model.py:445-476  (post_quant_conv, Decoder, last
    ``desired_length`` frames)
  * ``Decoder.forward``                        algorithms/vae/video_vae/model.py:255-281
  * ``ResnetBlock3D``, ``AttnBlock3D``, ``PaddedConv3D``, ``SpatialUpsample2x``, ``Spatial2xTime2x3DUpsample``
                                               algorithms/vae/common/modules/{resnet,attention,conv,updownsample}.py
with the default (causal) module choice of ``VideoVAE.__init__`` -- what ``VideoVAE_K600.ckpt`` is built from (bash/k600/*.sh:17).
The module registers the reference's decoder state-dict names (``decoder.*``, ``post_quant_conv.conv.*``), so the ``vae.``-prefixed
keys of a reference checkpoint load with ``load_state_dict``.  Every value is computed by a HIP kernel behind the C ABI: the 3x3x3
causal convolutions are three implicit-GEMM 3x3 convolutions (frames t-2, t-1, t) accumulated in fp32, the 1x1x1 projections and the
per-frame attention products are MFMA GEMMs, GroupNorm / upsampling / softmax are the HBM-bound kernels of csrc/vae.hip.
Host code only sequences the calls (the decode runs once per generated video, after the 50-step sampler).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import capi

BF = torch.bfloat16
_P = capi.ptr
_S = capi.stream_ptr


def _pad_to(n: int, m: int) -> int:
    return -(-n // m) * m


class VideoVAEDecoder(nn.Module):
    def __init__(self, z_channels: int = 16, hidden_size: int = 128, hidden_size_mult: Sequence[int] = (1, 2, 4, 4), num_res_blocks: int = 2,
                 embed_dim: Optional[int] = None, use_quant_layer: bool = True,
                 spatial_upsample: Sequence[str] = ("", "SpatialUpsample2x", "Spatial2xTime2x3DUpsample", "Spatial2xTime2x3DUpsample"),
                 attn_resolutions: Sequence[int] = (), is_causal: bool = True):
        super().__init__()
        if not is_causal:
            raise NotImplementedError("only the causal VideoVAE (VideoVAE's default, the K600 checkpoint) is supported")
        if tuple(attn_resolutions):
            raise NotImplementedError("decoder attention inside the up levels (attn_resolutions) is not supported; the mid attention is")
        for u in spatial_upsample:
            if u not in ("", "SpatialUpsample2x", "Spatial2xTime2x3DUpsample"):
                raise NotImplementedError(f"unsupported decoder upsample '{u}'")
        self.z, self.hidden, self.mult, self.nres = int(z_channels), int(hidden_size), tuple(hidden_size_mult), int(num_res_blocks)
        self.embed = int(embed_dim if embed_dim is not None else z_channels)
        self.use_quant, self.up_kind = bool(use_quant_layer), tuple(spatial_upsample)
        self.levels = len(self.mult)
        chans = [self.hidden * m for m in self.mult]
        for c in chans:
            if c not in (128, 256, 512, 1024):
                raise ValueError(f"decoder width {c} not in {{128, 256, 512, 1024}} (GroupNorm / GEMM tiling of the engine)")
        self._specs: List[Tuple[str, Tuple[int, ...]]] = []

        def conv3(name, ci, co, k=(3, 3, 3)):
            self._specs += [(f"{name}.conv.weight", (co, ci, *k)), (f"{name}.conv.bias", (co,))]

        def norm(name, c):
            self._specs += [(f"{name}.weight", (c,)), (f"{name}.bias", (c,))]

        def res(name, ci, co):
            norm(f"{name}.norm1", ci)
            conv3(f"{name}.conv1", ci, co)
            norm(f"{name}.norm2", co)
            conv3(f"{name}.conv2", co, co)
            if ci != co:
                conv3(f"{name}.nin_shortcut", ci, co, (1, 1, 1))
        if self.use_quant:
            conv3("post_quant_conv", self.embed, self.z, (1, 1, 1))
        top = chans[-1]
        conv3("decoder.conv_in", self.z, top)
        res("decoder.mid.block_1", top, top)
        norm("decoder.mid.attn_1.norm", top)
        for n in ("q", "k", "v", "proj_out"):
            conv3(f"decoder.mid.attn_1.{n}", top, top, (1, 1, 1))
        res("decoder.mid.block_2", top, top)
        self.plan: List[Tuple[int, List[Tuple[str, int, int]], str]] = []
        cin = top
        for lvl in reversed(range(self.levels)):
            blocks = []
            for i in range(self.nres + 1):
                blocks.append((f"decoder.up.{lvl}.block.{i}", cin, chans[lvl]))
                cin = chans[lvl]
            self.plan.append((lvl, blocks, self.up_kind[lvl]))
        # registration order of the reference: up modules are inserted at the front, so up.0 comes first in the state dict
        for lvl, blocks, kind in sorted(self.plan, key=lambda e: e[0]):
            for name, ci, co in blocks:
                res(name, ci, co)
            # both upsample modules hold a PaddedConv3D named `conv`, whose nn.Conv3d is `conv` again: ...upsample.conv.conv.weight
            if kind == "SpatialUpsample2x":
                conv3(f"decoder.up.{lvl}.upsample.conv", blocks[-1][2], blocks[-1][2], (1, 3, 3))
            elif kind == "Spatial2xTime2x3DUpsample":
                conv3(f"decoder.up.{lvl}.upsample.conv", blocks[-1][2], blocks[-1][2])
        norm("decoder.norm_out", chans[0])
        conv3("decoder.conv_out", chans[0], 3)
        self._names = [n for n, _ in self._specs]
        for name, shape in self._specs:
            *path, leaf = name.split(".")
            node: nn.Module = self
            for part in path:
                if part not in node._modules:
                    node.add_module(part, nn.Module())
                node = node._modules[part]
            node.register_parameter(leaf, nn.Parameter(torch.zeros(shape), requires_grad=False))
        self._packed: Dict[str, torch.Tensor] = {}
        self._sig = None

    # ------------------------------------------------------------------ weights -> bf16 GEMM operands
    def init_random(self, seed: int = 0) -> None:
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for n, p in self.named_parameters():
                if n.endswith("bias"):
                    v = 0.02 * torch.randn(p.shape, generator=g)
                elif p.ndim == 1:
                    v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
                else:
                    v = torch.randn(p.shape, generator=g) / (p[0].numel() ** 0.5)
                p.copy_(v.to(p.device))

    def _sync(self) -> None:
        params = dict(self.named_parameters())
        sig = tuple((t.data_ptr(), t._version) for t in params.values())
        if sig == self._sig:
            return
        pk: Dict[str, torch.Tensor] = {}
        for name, t in params.items():
            if not t.is_cuda:
                raise RuntimeError(f"parameter {name} is on {t.device}; move the module to the GPU first (there is no CPU path)")
            if not name.endswith(".conv.weight"):
                continue
            w = t.detach().float()
            co, ci, kt, kh, kw = w.shape
            cip, cop = _pad_to(ci, 64), _pad_to(co, 8)
            if (kh, kw) == (1, 1):  # 1x1x1: a Linear [co][ci]
                m = torch.zeros(cop, cip, device=w.device)
                m[:co, :ci] = w.view(co, ci)
                pk[name] = m.to(BF).contiguous()
            else:  # one packed [co][tap][ci] operand per temporal tap (dfot_op_pack_conv3: [Co][Ci][3][3] fp32 -> [Co][9][Ci] bf16)
                taps = []
                for dt in range(kt):
                    w2 = torch.zeros(cop, cip, 3, 3, device=w.device)
                    w2[:co, :ci] = w[:, :, dt]
                    out = torch.empty(cop, 9 * cip, dtype=BF, device=w.device)
                    capi.check(capi.lib.dfot_op_pack_conv3(_P(w2.contiguous()), _P(out), cop, cip, 0, _S()))
                    taps.append(out)
                pk[name] = torch.stack(taps)
            b = torch.zeros(cop, device=w.device)
            b[:co] = params[name[:-6] + "bias"].detach().float()
            pk[name[:-6] + "bias"] = b
        self._packed, self._sig = pk, sig

    # ------------------------------------------------------------------ building blocks (channels-last [B][T][H][W][C])
    def _gn(self, x: torch.Tensor, name: str, silu: bool, b: int) -> torch.Tensor:
        c = x.shape[-1]
        pixels = x.numel() // (b * c)
        out = torch.empty(x.shape, dtype=BF, device=x.device)
        scratch = torch.empty(int(capi.lib.dfot_op_groupnorm_scratch_floats(b, pixels)), device=x.device)
        p = dict(self.named_parameters())
        capi.check(capi.lib.dfot_op_groupnorm(_P(x), _P(p[name + ".weight"].detach()), _P(p[name + ".bias"].detach()), 1e-6, _P(out), _P(scratch), b,
                                              pixels, c, int(silu), _S()))
        return out

    def _conv(self, x: torch.Tensor, name: str, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
        """PaddedConv3D (causal, first-frame replication) on a bf16 [B][T][H][W][Ci] operand -> fp32 [B][T][H][W][Co] (+ resid)"""
        wp, bias = self._packed[name + ".conv.weight"], self._packed[name + ".conv.bias"]
        b, t, h, w, ci = x.shape
        if wp.ndim == 2:  # 1x1x1
            co = wp.shape[0]
            out = torch.empty(b, t, h, w, co, device=x.device)
            m = b * t * h * w
            capi.check(capi.lib.dfot_op_gemm_f32(_P(x), ci, _P(wp), _P(bias), _P(resid), _P(out), co, m, co, ci, _S()))
            return out
        kt, co = wp.shape[0], wp.shape[1]
        out = torch.empty(b, t, h, w, co, device=x.device)
        for dt in range(kt):
            shift = kt - 1 - dt
            xs = x
            if shift:
                xs = torch.empty_like(x)
                capi.check(capi.lib.dfot_op_frame_shift(_P(x), _P(xs), b, t, h * w * ci, shift, _S()))
            acc = resid if dt == 0 else out
            capi.check(capi.lib.dfot_op_conv3x3_f32(_P(xs), _P(wp[dt]), _P(bias) if dt == 0 else None, _P(acc), _P(out), b * t, h, w, ci, co, _S()))
        return out

    def _bf(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty(x.shape, dtype=BF, device=x.device)
        capi.check(capi.lib.dfot_op_f32_to_bf16(_P(x), _P(out), x.numel(), _S()))
        return out

    def _res(self, x: torch.Tensor, name: str, ci: int, co: int, b: int) -> torch.Tensor:
        h = self._conv(self._gn(x, name + ".norm1", True, b), name + ".conv1")
        sc = x if ci == co else self._conv(self._bf(x), name + ".nin_shortcut")
        return self._conv(self._gn(h, name + ".norm2", True, b), name + ".conv2", resid=sc)

    def _attn(self, x: torch.Tensor, name: str, b: int) -> torch.Tensor:
        """AttnBlock3D: per frame, one head over the H*W positions with all C channels"""
        bb, t, h, w, c = x.shape
        n = h * w
        hn = self._gn(x, name + ".norm", False, b)
        q, k, v = (torch.empty(bb * t * n, c, dtype=BF, device=x.device) for _ in range(3))
        for dst, nm in ((q, "q"), (k, "k"), (v, "v")):
            capi.check(capi.lib.dfot_op_gemm_bf16(_P(hn), c, _P(self._packed[f"{name}.{nm}.conv.weight"]), _P(self._packed[f"{name}.{nm}.conv.bias"]),
                                                  _P(dst), c, bb * t * n, c, c, _S()))
        o = torch.empty(bb * t * n, c, dtype=BF, device=x.device)
        scores = torch.empty(n, n, device=x.device)
        probs = torch.empty(n, n, dtype=BF, device=x.device)
        vt = torch.empty(c, n, dtype=BF, device=x.device)
        for f in range(bb * t):
            qf, kf, vf, of = (a[f * n:(f + 1) * n] for a in (q, k, v, o))
            capi.check(capi.lib.dfot_op_gemm_f32(_P(qf), c, _P(kf), None, None, _P(scores), n, n, n, c, _S()))          # S = Q K^T
            capi.check(capi.lib.dfot_op_softmax_rows(_P(scores), _P(probs), n, n, float(c) ** -0.5, _S()))
            capi.check(capi.lib.dfot_op_transpose_bf16(_P(vf), _P(vt), n, c, _S()))                                       # V^T [C][N]
            capi.check(capi.lib.dfot_op_gemm_bf16(_P(probs), n, _P(vt), None, _P(of), c, n, c, n, _S()))                  # O = P V
        out = torch.empty_like(x)
        capi.check(capi.lib.dfot_op_gemm_f32(_P(o), c, _P(self._packed[f"{name}.proj_out.conv.weight"]), _P(self._packed[f"{name}.proj_out.conv.bias"]),
                                             _P(x), _P(out), c, bb * t * n, c, c, _S()))
        return out

    # ------------------------------------------------------------------ decode
    @torch.no_grad()
    def decode(self, z: torch.Tensor, desired_length: Optional[int] = None) -> torch.Tensor:
        """VideoVAE.decode: z (B, z_channels, T, H, W) -> (B, 3, T', 8H, 8W) with T' = 1 + 4 (T - 1) for the default module choice; the
        LAST ``desired_length`` frames are returned when given."""
        if z.ndim != 5 or z.shape[1] != (self.embed if self.use_quant else self.z):
            raise ValueError(f"z has shape {tuple(z.shape)}, expected (B, {self.embed if self.use_quant else self.z}, T, H, W)")
        self._sync()
        dev = next(self.parameters()).device
        capi.require_device(dev, z=z)
        b, cz, t, h, w = z.shape
        if (t * h * w) % 128:
            raise ValueError(f"T*H*W = {t * h * w} must be a multiple of 128 (GEMM row tiles)")
        cl = torch.zeros(b, t, h, w, _pad_to(cz, 64), device=dev)
        cl[..., :cz] = z.detach().float().permute(0, 2, 3, 4, 1)
        x = self._bf(cl)
        if self.use_quant:
            y = self._conv(x, "post_quant_conv")                      # [.., pad8(z)] fp32
            cl = torch.zeros(b, t, h, w, _pad_to(self.z, 64), device=dev)
            cl[..., : self.z] = y[..., : self.z]
            x = self._bf(cl)
        hcur = self._conv(x, "decoder.conv_in")
        top = self.hidden * self.mult[-1]
        hcur = self._res(hcur, "decoder.mid.block_1", top, top, b)
        hcur = self._attn(hcur, "decoder.mid.attn_1", b)
        hcur = self._res(hcur, "decoder.mid.block_2", top, top, b)
        for lvl, blocks, kind in self.plan:
            for name, ci, co in blocks:
                hcur = self._res(hcur, name, ci, co, b)
            if kind:
                bb, tt, hh, ww, cc = hcur.shape
                mode = 1 if kind == "Spatial2xTime2x3DUpsample" else 0
                t2 = 1 + 2 * (tt - 1) if mode == 1 else tt
                up = torch.empty(bb, t2, 2 * hh, 2 * ww, cc, device=dev)
                capi.check(capi.lib.dfot_op_upsample3d(_P(hcur), _P(up), bb, tt, hh, ww, cc, mode, _S()))
                hcur = self._conv(self._bf(up), f"decoder.up.{lvl}.upsample.conv")
        y = self._conv(self._gn(hcur, "decoder.norm_out", True, b), "decoder.conv_out")[..., :3]
        out = y.permute(0, 4, 1, 2, 3).contiguous()
        if desired_length is not None:
            out = out[:, :, -desired_length:]
            assert out.shape[2] == desired_length, f"Desired length {desired_length} does not match decoded length {out.shape[2]}"
        return out

    def load_reference_state_dict(self, state_dict: Dict[str, torch.Tensor]) -> List[str]:
        """keys of a reference VideoVAE (``decoder.*``, ``post_quant_conv.*``; optionally ``vae.``-prefixed as in the Lightning checkpoint,
        video_vae/model.py:520-527); encoder / quant_conv / loss keys are ignored and returned.  Strict on the decoder's own keys."""
        own = dict(self.named_parameters())
        ignored, seen = [], set()
        with torch.no_grad():
            for k, v in state_dict.items():
                n = k[4:] if k.startswith("vae.") else k
                if n in own:
                    if tuple(v.shape) != tuple(own[n].shape):
                        raise ValueError(f"size mismatch for {n}: {tuple(v.shape)} vs {tuple(own[n].shape)}")
                    own[n].copy_(v)
                    seen.add(n)
                else:
                    ignored.append(k)
        missing = [n for n in own if n not in seen]
        if missing:
            raise ValueError(f"keys not found in the checkpoint: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        return ignored


@torch.no_grad()
def decode_latents(vae: VideoVAEDecoder, latents: torch.Tensor, n_frames: int, vae_batch_size: int = 2, shape: str = "b t c h w") -> torch.Tensor:
    """``BaseVideoAlgo._decode`` for a VideoVAE (base_pytorch_video_algo.py:553-629): latents in the sampler's ``b t c h w`` layout, chunks of
    ``vae.batch_size`` videos, ``decode(y, n_frames) * 0.5 + 0.5``, result back in ``b t c h w`` (frames in [0, 1])."""
    if shape != "b t c h w":
        raise ValueError("only the 'b t c h w' layout of the sampling path is supported")
    x = latents.permute(0, 2, 1, 3, 4)
    n_chunks = (x.shape[0] + vae_batch_size - 1) // vae_batch_size
    outs = [vae.decode(ch, n_frames) * 0.5 + 0.5 for ch in torch.chunk(x, n_chunks, 0)]
    return torch.cat(outs, 0).permute(0, 2, 1, 3, 4).contiguous()
