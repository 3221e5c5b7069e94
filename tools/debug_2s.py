"""diagnostic: run-to-run differences of the backbone forward under the two-stream block schedule (DFOT_UVIT_TWO_STREAM / set_option):
which workspace buffer of the level-3 blocks differs between two runs of the same forward"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dfot_amd
from dfot_amd import capi
from oracle import pose as opose, uvit as ouvit
res = int(os.environ.get("RES", "64"))
mid = int(os.environ.get("MID", "20"))
ocfg = ouvit.UViTConfig(resolution=res, num_mid_blocks=mid)
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
m3 = 2 * 8 * (res // 16) ** 2
c3 = ocfg.channels[3]
raws = {"raw_s1": (m3 * c3 // 2, torch.bfloat16), "raw_q": (m3 * c3 // 2, torch.bfloat16), "raw_cat": (m3 * 5 * c3 // 2, torch.bfloat16),
        "raw_part": (3 * m3 * c3, torch.float32), "raw_x3": (m3 * c3, torch.float32)}

def grab():
    out = {}
    for n, (nf, dt) in raws.items():
        buf = torch.empty(nf, device="cuda", dtype=torch.float32)
        capi.check(capi.lib.dfot_uvit_read_tap(model._handle, n.encode(), capi.ptr(buf), buf.numel(), capi.stream_ptr()))
        out[n] = buf.view(dt).clone()
    if not os.environ.get("NOSYNC"):
        torch.cuda.synchronize()
    return out

model.set_option("debug_stop_after_mid", 1)
with torch.no_grad():
    for ts in (0, int(os.environ.get("TS", "1"))):
        model.set_option("two_stream", ts)
        runs = []
        for _ in range(int(os.environ.get("RUNS", "4"))):
            o = model(x, k, cond, None).clone()
            r = grab() if not os.environ.get("NOGRAB") else {}
            r["out"] = o
            runs.append(r)
        for n in runs[0]:
            a = runs[0][n].float()
            if n == "raw_cat":
                a2 = a.view(m3, 5 * c3)
                for nm, sl in (("cat[:, :C] (attention out)", slice(0, c3)), ("cat[:, C:] (SiLU mlp_h)", slice(c3, 5 * c3))):
                    d = [int((r[n].float().view(m3, 5 * c3)[:, sl] != a2[:, sl]).sum()) for r in runs[1:]]
                    print(f"two_stream={ts} {nm}: elements differing from run 0: {d}")
            elif n == "raw_part":
                a3 = a.view(3, m3, c3)
                for i in range(3):
                    d = [int((r[n].float().view(3, m3, c3)[i] != a3[i]).sum()) for r in runs[1:]]
                    print(f"two_stream={ts} out_part slice {i}: elements differing from run 0: {d}")
            else:
                d = [int((r[n].float() != a).sum()) for r in runs[1:]]
                print(f"two_stream={ts} {n}: elements differing from run 0: {d} of {a.numel()}")
                if n == "raw_q" and ts and any(d):
                    for r in runs[1:]:
                        df = (r[n].float() != a).nonzero().flatten()
                        if df.numel():
                            ntok = 8 * (res // 16) ** 2
                            for i in df[:16].tolist():
                                b_, rem = divmod(i, 9 * ntok * 128)
                                hd, rem = divmod(rem, ntok * 128)
                                tok, e = divmod(rem, 128)
                                print(f"   q[b={b_}, head={hd}, tok={tok}, e={e}] = {float(a[i]):.6f} vs {float(r[n].float()[i]):.6f}")
                            break
                if n == "raw_x3" and ts and any(d):
                    for r in runs[1:]:
                        df = (r[n].float() != a).view(m3, c3)
                        if df.any():
                            rows = df.any(1).nonzero().flatten()
                            cols = df.any(0).nonzero().flatten()
                            print(f"   x3 rows differing: {rows.numel()} in [{int(rows.min())}, {int(rows.max())}], cols {cols.numel()} in [{int(cols.min())}, {int(cols.max())}]; "
                                  f"max abs diff {float((r[n].float() - a).abs().max()):.3e}")
                            break
