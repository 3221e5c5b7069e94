"""diagnostic: run-to-run differences of the backbone forward under the two-stream block schedule"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dfot_amd
from oracle import pose as opose, uvit as ouvit
res = int(os.environ.get("RES", "64"))
ocfg = ouvit.UViTConfig(resolution=res)
params = ouvit.seeded_params(ocfg, 3)
cfg = dict(channels=list(ocfg.channels), emb_channels=ocfg.emb_channels, patch_size=2, block_types=list(ocfg.block_types),
           num_updown_blocks=list(ocfg.num_updown_blocks), num_mid_blocks=ocfg.num_mid_blocks, num_heads=ocfg.num_heads, pos_emb_type="rope",
           use_fourier_noise_embedding=True, conditioning=dict(dim=180))
model = dfot_amd.UViT3DPose(cfg, x_shape=(3, res, res), max_tokens=8).cuda()
model.load_state_dict(params, strict=True)
g = torch.Generator().manual_seed(0)
x = torch.randn(2, 8, 3, res, res, generator=g).cuda()
k = torch.randn(2, 8, generator=g).cuda()
pz = torch.eye(3, 4).reshape(1, 1, 12).repeat(2, 8, 1)
pz[..., 3] = torch.linspace(0, 0.5, 8)
cond = opose.ray_encoding(torch.cat([torch.tensor([0.5, 0.9, 0.5, 0.5]).repeat(2, 8, 1), pz], -1), res).cuda()
ctx = torch.cuda.stream(torch.cuda.Stream()) if os.environ.get("USER_STREAM") else torch.no_grad()
with torch.no_grad(), ctx:
    outs = [model(x, k, cond, None).clone() for _ in range(4)]
    taps = {}
    for ts in (0, int(os.environ.get("DFOT_UVIT_TWO_STREAM", "1"))):
        model.set_option("two_stream", ts)
        runs = []
        for _ in range(3):
            model(x, k, cond, None)
            runs.append({n: model.read_tap(n, c, l, 2).clone() for n, c, l in (("down2", ocfg.channels[3], 3), ("mid", ocfg.channels[3], 3), ("up2", ocfg.channels[2], 2))})
        for n in runs[0]:
            d = [float((r[n] - runs[0][n]).abs().max()) for r in runs[1:]]
            print(f"two_stream={ts} tap {n}: max |run_i - run_0| = {d}, scale {float(runs[0][n].abs().max()):.3f}")
torch.cuda.synchronize()
for i in range(1, 4):
    d = (outs[i] - outs[0]).abs()
    print(f"run {i} vs run 0: max abs diff {float(d.max()):.3e}, elements differing {int((d > 0).sum())} of {d.numel()}, out scale {float(outs[0].abs().max()):.3f}")
