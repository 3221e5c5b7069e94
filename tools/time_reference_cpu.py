#!/usr/bin/env python3
"""Time the reference's OWN source (through tools/ref_loader.py) and this repo's CPU oracle on the same inputs and weights,
same thread count -- BASELINE.md section 3.  Runs only in the build container (needs /root/reference).

  python tools/time_reference_cpu.py [threads]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_loader  # noqa: E402
import make_golden as MG  # noqa: E402
import make_golden_dit as MD  # noqa: E402
from oracle import dit as odit, pose as opose, uvit as ouvit  # noqa: E402


def timed(fn, reps=1):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


@torch.no_grad()
def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    torch.set_num_threads(threads)
    R = ref_loader.install()
    A = R["AttrDict"]
    print(f"threads = {threads}")
    # RE10K backbone, B=1, T=8, 256x256 (one window-forward)
    cfg = MG.algo_cfg(A, 256, MG.W64)
    algo, ocfg, params = MG.build_algo(R, cfg)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 8, 3, 256, 256, generator=g)
    k = 0.125 * algo.diffusion_model.logsnr[torch.randint(0, 1000, (1, 8), generator=g)]
    cond = opose.ray_encoding(MG.synth_poses(1, 8, seed=1), 256)
    t_ref, a = timed(lambda: algo.diffusion_model.model(x, k, cond, None))
    t_orc, b = timed(lambda: ouvit.forward(params, ocfg, x, k, cond, None))
    print(f"UViT3DPose window-forward (1x8x3x256x256): reference source {t_ref:.1f} s, oracle {t_orc:.1f} s, "
          f"max |diff| {float((a - b).abs().max()):.2e}")
    del algo, params
    # DiT/XL, one video
    dcfg = odit.DiTConfig()
    m, p = MD.ref_dit(R, dcfg, 0)
    x = torch.randn(1, 5, 16, 16, 16, generator=g)
    kk = torch.randint(0, 1000, (1, 5), generator=g)
    t_ref, a = timed(lambda: m(x, kk), 2)
    t_orc, b = timed(lambda: odit.forward(p, dcfg, x, kk), 2)
    print(f"DiT3D @DiT/XL forward (1x5x16x16x16): reference source {t_ref:.2f} s, oracle {t_orc:.2f} s, max |diff| {float((a - b).abs().max()):.2e}")
    del m, p
    # DifferenceDiT3D bash/k600, one video
    fcfg = odit.DiffDiTConfig()
    m, p = MD.ref_diffdit(R, fcfg, 0)
    x = torch.randn(1, 10, 16, 16, 16, generator=g)
    kk = torch.randint(0, 1000, (1, 10), generator=g)
    t_ref, a = timed(lambda: m(x, kk))
    t_orc, b = timed(lambda: odit.diff_forward(p, fcfg, x, kk))
    print(f"DifferenceDiT3D XL-64-1 forward (1x10x16x16x16): reference source {t_ref:.2f} s, oracle {t_orc:.2f} s, "
          f"max |diff| {float((a - b).abs().max()):.2e}")


if __name__ == "__main__":
    main()
