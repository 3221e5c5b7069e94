"""CPU: the oracle's restatement of the VideoVAE decode path and the product module's state-dict inventory against the fixture captured
from the reference's own VideoVAE source (tests/golden/vae_decode.npz, tools/make_golden.py ONLY=vae)."""
import ast
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import vae as ovae


def _load():
    g = np.load(os.path.join(GOLDEN, "vae_decode.npz"))
    shapes = {str(n): ast.literal_eval(str(s)) for n, s in zip(g["names"], g["shapes"])}
    return g, shapes


def test_oracle_vae_decode_vs_reference_fixture():
    g, shapes = _load()
    p = {n: ovae.seeded_tensor(n, s) for n, s in shapes.items()}
    z = torch.from_numpy(g["z"])[:1]   # one of the two videos keeps the CPU test short
    with torch.no_grad():
        out = ovae.decode(p, ovae.VAEConfig(), z)
    ref = torch.from_numpy(g["frames"])[:1]
    assert out.shape == ref.shape
    torch.testing.assert_close(out, ref, rtol=1e-3, atol=2e-4)
    with torch.no_grad():
        fr = ovae.decode_latents(p, ovae.VAEConfig(), z.permute(0, 2, 1, 3, 4), 7)
    torch.testing.assert_close(fr, (ref[:, :, -7:] * 0.5 + 0.5).permute(0, 2, 1, 3, 4), rtol=1e-3, atol=2e-4)


def test_product_decoder_registers_the_reference_keys():
    """the drop-in module's parameter names / shapes == the reference VideoVAE's decoder + post_quant_conv state dict"""
    import dfot_amd
    _, shapes = _load()
    dec = dfot_amd.VideoVAEDecoder(z_channels=16, hidden_size=128, embed_dim=16)   # constructing it does not touch the GPU
    own = {n: tuple(t.shape) for n, t in dec.named_parameters()}
    assert own == shapes
