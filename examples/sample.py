#!/usr/bin/env python3
"""End-to-end sampling with the MI355X engine, the way the reference's validation loop calls its algorithm.

  python examples/sample.py re10k    [--ckpt DFoT_RE10K.ckpt] [--frames 8] [--steps 50] [--out out.npz]
  python examples/sample.py k600     [--ckpt K600.ckpt] [--batch 8]
  python examples/sample.py k600diff [--ckpt ...]

Without --ckpt the backbone gets seeded random weights (there is no network here to fetch the released checkpoints);
with it, the reference's .ckpt / ema.safetensors is read by dfot_amd.load_reference_checkpoint (keys
`diffusion_model.model.*`, optional `_orig_mod.` prefix, EMA weights).  Inputs are synthetic unless --inputs points at an
.npz with `xs` (B,T,C,H,W, already normalised) and, for re10k, `conditions` (B,T,16 raw camera poses).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dfot_amd  # noqa: E402
from bench import RE10K, synth_poses  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", choices=["re10k", "k600", "k600diff"])
    ap.add_argument("--ckpt")
    ap.add_argument("--inputs")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="sample_out.npz")
    a = ap.parse_args()
    gen = torch.Generator(device="cuda").manual_seed(a.seed)
    noise = dfot_amd.device_noise_fn(gen)
    conds = None
    if a.model == "re10k":
        model = dfot_amd.UViT3DPose(RE10K, x_shape=(3, 256, 256), max_tokens=8).cuda()
        long_video = a.frames > 8
        cfg = dfot_amd.SamplerConfig(
            x_shape=(3, 256, 256), max_tokens=8, diffusion=dfot_amd.DiffusionConfig(sampling_timesteps=a.steps),
            prediction_guidance=(dict(name="stabilized_vanilla", guidance_scale=4.0, stabilization_level=0.02) if long_video
                                 else dict(name="vanilla", guidance_scale=4.0)),
            interpolation_guidance=dict(name="vanilla", guidance_scale=1.5), keyframe_density=(0.0625 if a.frames >= 128 else 0.5) if long_video else None,  # >= max_tokens key frames
            interpolation_max_batch_size=4)
        sampler = dfot_amd.DFoTVideoPoseSampler(cfg, model, noise)
        xs = torch.randn(a.batch, a.frames, 3, 256, 256, generator=torch.Generator().manual_seed(a.seed))
        conds = synth_poses(a.batch, a.frames, 100 + a.seed)
        n_ctx = 1
    else:
        diff = a.model == "k600diff"
        if diff:
            bb = dict(name="difference_dit3d", variant="factorized_matrix_attention", pos_emb_type="sinusoidal_2d", merge_type="interleaved",
                      patch_size=1, embed_col_dim=64, embed_row_dim=1152, num_heads=12, num_col_heads=1, num_row_heads=16, depth=28,
                      mlp_ratio=4.0, spatial_mlp_ratio=4.0, use_bias=True, matrix_block="matrix")
            model = dfot_amd.DifferenceDiT3D(bb, x_shape=(16, 16, 16), max_tokens=5).cuda()
        else:
            bb = dict(name="dit3d", variant="full", pos_emb_type="rope_3d", patch_size=1, hidden_size=1152, depth=28, num_heads=16)
            model = dfot_amd.DiT3D(bb, x_shape=(16, 16, 16), max_tokens=5).cuda()
        cfg = dfot_amd.SamplerConfig(x_shape=(16, 16, 16), max_tokens=10 if diff else 5,
                                     diffusion=dfot_amd.DiffusionConfig(sampling_timesteps=a.steps, beta_schedule="cosine", is_continuous=False))
        sampler = (dfot_amd.DifferenceDFoTVideoSampler if diff else dfot_amd.DFoTVideoSampler)(cfg, model, noise)
        xs = torch.randn(a.batch, 5, 16, 16, 16, generator=torch.Generator().manual_seed(a.seed))
        n_ctx = 2
    if a.ckpt:
        ignored = dfot_amd.load_reference_checkpoint(model, a.ckpt)
        print(f"loaded {a.ckpt} ({len(ignored)} non-backbone keys ignored)")
    else:
        model.init_random(seed=a.seed)
    if a.inputs:
        data = np.load(a.inputs)
        xs = torch.from_numpy(data["xs"]).float()
        conds = torch.from_numpy(data["conditions"]).float() if "conditions" in data.files else conds
    xs = xs.cuda()
    conds = None if conds is None else conds.cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if a.model == "k600diff":
        out = sampler._sample_all_videos(xs, n_context_tokens=n_ctx)["prediction"]
    else:
        out = sampler._predict_videos(xs, n_context_tokens=n_ctx, conditions=conds)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gen_frames = (out.shape[1] - n_ctx) * out.shape[0]
    print(f"{a.model}: {tuple(out.shape)} in {dt:.2f} s  ({gen_frames / dt:.2f} generated frames/s, "
          f"{sampler.window_forwards} backbone forwards of one window)")
    np.savez_compressed(a.out, out=out.cpu().numpy())


if __name__ == "__main__":
    main()
