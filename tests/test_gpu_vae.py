"""GPU parity of the VideoVAE decode path (latents -> frames for the Kinetics-600 configuration; algorithms/common/
base_pytorch_video_algo.py:553-629, algorithms/vae/video_vae/model.py) against the fixture captured from the reference's own VideoVAE
source (tests/golden/vae_decode.npz) and, op by op, against plain PyTorch.  Tolerance: rel-L2 <= 2e-2 on the decoded frames (bf16 GEMM
operands, fp32 accumulation / streams / statistics; ~45 convolutions deep)."""
import ast
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def P(t):
    return C.c_void_p(t.data_ptr())


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


@pytest.fixture(scope="module")
def capi():
    import dfot_amd  # noqa: F401
    from dfot_amd import capi as c
    return c


def test_vae_pieces_vs_torch(capi):
    g = torch.Generator().manual_seed(3)
    b, t, h, w, c = 2, 3, 8, 4, 128
    x = torch.randn(b, c, t, h, w, generator=g)
    cl = x.permute(0, 2, 3, 4, 1).contiguous().cuda()
    # trilinear causal upsample (Spatial2xTime2x3DUpsample) and nearest (SpatialUpsample2x)
    ref = torch.cat([F.interpolate(x[:, :, :1], scale_factor=(1, 2, 2), mode="trilinear"),
                     F.interpolate(x[:, :, 1:], scale_factor=(2, 2, 2), mode="trilinear")], 2)
    out = torch.empty(b, 1 + 2 * (t - 1), 2 * h, 2 * w, c, device="cuda")
    capi.check(capi.lib.dfot_op_upsample3d(P(cl), P(out), b, t, h, w, c, 1, S()))
    torch.testing.assert_close(out.cpu().permute(0, 4, 1, 2, 3), ref, rtol=1e-5, atol=1e-5)
    one = torch.empty(b, 1, 2 * h, 2 * w, c, device="cuda")   # a single latent frame: spatial only
    capi.check(capi.lib.dfot_op_upsample3d(P(cl[:, :1].contiguous()), P(one), b, 1, h, w, c, 1, S()))
    torch.testing.assert_close(one.cpu().permute(0, 4, 1, 2, 3), F.interpolate(x[:, :, :1], scale_factor=(1, 2, 2), mode="trilinear"), rtol=1e-5, atol=1e-5)
    near = torch.empty(b, t, 2 * h, 2 * w, c, device="cuda")
    capi.check(capi.lib.dfot_op_upsample3d(P(cl), P(near), b, t, h, w, c, 0, S()))
    torch.testing.assert_close(near.cpu().permute(0, 4, 1, 2, 3), F.interpolate(x, scale_factor=(1, 2, 2), mode="nearest"))
    # GroupNorm over (T, H, W) with and without SiLU
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    scratch = torch.empty(int(capi.lib.dfot_op_groupnorm_scratch_floats(b, t * h * w)), device="cuda")
    gd, bd = gamma.cuda(), beta.cuda()
    for silu in (0, 1):
        o = torch.empty(b, t, h, w, c, device="cuda", dtype=torch.bfloat16)
        capi.check(capi.lib.dfot_op_groupnorm(P(cl), P(gd), P(bd), 1e-6, P(o), P(scratch), b, t * h * w, c, silu, S()))
        r = F.group_norm(x, 32, gamma, beta, eps=1e-6)
        r = r * torch.sigmoid(r) if silu else r
        assert rel(o.float().cpu().permute(0, 4, 1, 2, 3), r) < 4e-3
    # causal frame shift
    xb = cl.to(torch.bfloat16)
    sh = torch.empty_like(xb)
    capi.check(capi.lib.dfot_op_frame_shift(P(xb), P(sh), b, t, h * w * c, 2, S()))
    assert torch.equal(sh[:, 2], xb[:, 0]) and torch.equal(sh[:, 0], xb[:, 0]) and torch.equal(sh[:, 1], xb[:, 0])
    # row softmax
    s = torch.randn(256, 384, generator=g) * 3
    sd = s.cuda()
    pr = torch.empty(256, 384, device="cuda", dtype=torch.bfloat16)
    capi.check(capi.lib.dfot_op_softmax_rows(P(sd), P(pr), 256, 384, 0.37, S()))
    assert rel(pr.float().cpu(), torch.softmax(s * 0.37, -1)) < 4e-3


def test_vae_decode_vs_reference_fixture():
    import dfot_amd
    from oracle import vae as ovae
    g = np.load(os.path.join(GOLDEN, "vae_decode.npz"))
    shapes = {str(n): ast.literal_eval(str(s)) for n, s in zip(g["names"], g["shapes"])}
    dec = dfot_amd.VideoVAEDecoder(z_channels=16, hidden_size=128, embed_dim=16).cuda()
    sd = {n: ovae.seeded_tensor(n, s) for n, s in shapes.items()}
    sd["vae.encoder.conv_in.weight"] = torch.zeros(1)      # a Lightning-style key the decoder ignores
    ignored = dec.load_reference_state_dict({("vae." + n if not n.startswith("vae.") else n): v for n, v in sd.items()})
    assert ignored == ["vae.encoder.conv_in.weight"]
    z = torch.from_numpy(g["z"]).cuda()
    out = dec.decode(z).cpu()
    ref = torch.from_numpy(g["frames"])
    r = rel(out, ref)
    print(f"VideoVAE decode (2 x 16 x 3 x 16 x 8 latents -> 9 frames of 128 x 64): rel-L2 vs the reference fixture {r:.3e}, "
          f"max_abs {(out - ref).abs().max().item():.3e}")
    assert out.shape == ref.shape and torch.isfinite(out).all() and r < 2e-2
    # the sampler-level convention (_decode): b t c h w latents, last n frames, [-1, 1] -> [0, 1], chunks of vae.batch_size videos
    fr = dfot_amd.decode_latents(dec, z.permute(0, 2, 1, 3, 4), n_frames=7, vae_batch_size=1).cpu()
    assert fr.shape == (2, 7, 3, 128, 64)
    assert rel(fr, (ref[:, :, -7:] * 0.5 + 0.5).permute(0, 2, 1, 3, 4)) < 2e-2
    with pytest.raises(ValueError):
        dec.decode(z.cpu())                                   # host latents are refused before any launch
    with pytest.raises(ValueError):
        dec.load_reference_state_dict({"decoder.conv_in.conv.weight": torch.zeros(1)})
