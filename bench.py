#!/usr/bin/env python3
"""DFoT RE10K sampling benchmark on MI355X (BASELINE.json metric: denoised frames/sec).

One "step" = one complete 8-frame sample of BASELINE config 2 (DFoT_RE10K: UViT3DPose 458.8 M params,
x_shape [3,256,256], context 1 frame, 50 DDIM steps, vanilla history guidance 4.0 => 100 window-forwards,
7 generated frames) on synthetic latents and seeded random-init weights.  With --gpus N every rank samples
its own video (independent units, no data-path collective; weak scaling); value = frames of all ranks / max time.

Also reports on rank 0:
  roofline     dominant kernel = level-2 flash attention (N=8192, d=64, 18 (batch,head) units per launch);
               algorithmic FLOPs per launch = 4*N^2*d*heads*Bm = 309.2 GFLOP, duration = HIP events
               recorded on the launch stream around every such launch inside the timed region.
  cpu_baseline the CPU oracle (oracle/, a plain-PyTorch fp32 port of the reference path) timed on the host cores on a
               bounded sample (one window-forward), converted to frames/s of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RE10K = dict(channels=[128, 256, 576, 1152], emb_channels=1024, patch_size=2,
             block_types=["ResBlock", "ResBlock", "TransformerBlock", "TransformerBlock"],
             num_updown_blocks=[3, 3, 6], num_mid_blocks=20, num_heads=9, pos_emb_type="rope",
             use_fourier_noise_embedding=True, conditioning=dict(dim=180))
WINDOW_FLOP = 6.63e12  # per 8-frame window-forward
GRAPH_DEFAULT = True  # sampler mode of the headline line: the north_star's hipGraph-captured step loop (`--eager` for the A/B)
# the reference's own source timed on its CPU path in the build container (tools/time_reference_cpu.py): it cannot
# travel to the GPU box, so the figure is quoted next to the oracle ("port") timing taken on the box itself
REFERENCE_SOURCE_CPU = {"seconds_per_window_forward": 20.9, "threads": 8, "host": "build container (8 vCPU Xeon 2.1 GHz), torch 2.10 CPU, fp32",
                        "how": "tools/time_reference_cpu.py: UViT3DPose.forward of the reference's source, Bm=1, T=8, 256x256"}


def executed_flop_fraction(res: int, frames: int, dead_rows: float, frozen_rows: float) -> float:
    """Share of the reference's per-forward FLOPs (WINDOW_FLOP) the engine executes in a sampling step.  Removed,
    all bit-exact: the pose patch-embed and every per-pixel FiLM projection (once per window instead of once per
    forward), the frame-local up path of frames whose output the composition discards (`dead_rows` = mean fraction of frame rows),
    the frame-local down path of frozen context frames (`frozen_rows`).  GFLOP per frame at 256x256, scaled by the pixel count."""
    px = (res / 256.0) ** 2
    total = 828.2 * px
    film = (24.2 + 51.5 + 25.8 + 29.0 + 24.2) * px         # pose embed + emb_layer of L0 / L1 ResBlocks, L2 / L3 transformer blocks
    frame_local = (3 * 9.66 + 2.42 + 3 * 9.66 + 2.72) * px  # three ResBlocks (convolutions only) + one between-level conv, levels 0 and 1
    return (total - film - frame_local * (dead_rows + frozen_rows)) / total


def synth_poses(b: int, t: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    k = torch.tensor([0.5, 0.9, 0.5, 0.5]).repeat(b, t, 1)
    ang = 0.1 * torch.randn(b, t, generator=g).cumsum(1)
    c, s, o, z = ang.cos(), ang.sin(), torch.ones_like(ang), torch.zeros_like(ang)
    rot = torch.stack([c, z, s, z, o, z, -s, z, c], -1).view(b, t, 3, 3)
    trans = torch.stack([torch.linspace(0, 0.5, t).repeat(b, 1), torch.zeros(b, t), torch.linspace(0, -0.3, t).repeat(b, 1)], -1)
    return torch.cat([k, torch.cat([rot, trans[..., None]], -1).reshape(b, t, 12)], -1)


def cpu_baseline(res: int, sample_forwards: int, frames: int, forwards_per_sample: int):
    from oracle import pose as opose, uvit as ouvit
    ocfg = ouvit.UViTConfig(resolution=res)
    params = ouvit.seeded_params(ocfg, 0)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 8, 3, res, res, generator=g)
    k = torch.randn(1, 8, generator=g)
    cond = opose.ray_encoding(synth_poses(1, 8, 7), res)
    cores = min(16, os.cpu_count() or 1)  # a 1-GPU box has a 16-core CPU share
    torch.set_num_threads(cores)
    ouvit.USE_SDPA = True  # the reference's own attention call (F.scaled_dot_product_attention); the explicit-softmax form
    #                        the parity tests use is 2.4x slower on a CPU (tools/time_reference_cpu.py)
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(sample_forwards):
            ouvit.forward(params, ocfg, x, k, cond, None)
    dt = (time.perf_counter() - t0) / sample_forwards
    return {"value": frames / (forwards_per_sample * dt), "unit": "frames/s", "cores": cores, "kind": "port",
            "reference_source": dict(REFERENCE_SOURCE_CPU, frames_per_s=frames / (forwards_per_sample * REFERENCE_SOURCE_CPU["seconds_per_window_forward"])),
            "sample": f"{sample_forwards} window-forward(s) of the oracle (Bm=1, T=8, {res}x{res}), {dt:.2f} s each; "
                      f"scaled to {forwards_per_sample} window-forwards per {frames}-frame sample"}


def k600_traffic(diff: bool, batch: int):
    """HBM bytes per launch of the timed attention kernel from the committed PMC passes (8 videos per launch)."""
    pmc = os.path.join(ROOT, "profiles", "r01_g_pmc_hbm_traffic_%s.json" % ("k600diff" if diff else "k600"))
    if batch != 8 or not os.path.exists(pmc):
        return None
    ks = [v for n, v in json.load(open(pmc))["kernels"].items() if n.startswith("attn_kernel_v2<128")]
    return ks[0]["hbm_bytes_per_launch_corrected"] if ks else None


def bench_train_k600(args, rank, world, dist):
    """BASELINE config 5's pattern on the model that has a training path (README @DiT/XL, K600 latents): one step = per-token
    independent noise levels -> noised forward -> fused-min-SNR v-loss -> hand-written backward -> all-reduce of the flat gradient
    buffer over the ranks (RCCL) -> global-norm clip + AdamW -> bf16 weight refresh.  `--batch` videos per GPU (weak scaling)."""
    import dfot_amd
    from dfot_amd import DifferenceDiT3D, DiT3D, DiT3DTrainer
    diff = args.workload == "train_k600diff"  # the model bash/k600/*.sh train: DifferenceDiT3D factorized matrix attention XL-64-1
    if diff:
        xl = dict(name="difference_dit3d", variant="factorized_matrix_attention", pos_emb_type="sinusoidal_2d", merge_type="interleaved",
                  patch_size=1, hidden_size=None, embed_col_dim=64, embed_row_dim=1152, num_heads=12, num_col_heads=1, num_row_heads=16,
                  depth=28, mlp_ratio=4.0, spatial_mlp_ratio=4.0, use_bias=True, matrix_block="matrix")
        init = DifferenceDiT3D(xl, x_shape=(16, 16, 16), max_tokens=5)
    else:
        xl = dict(name="dit3d", variant="full", pos_emb_type="rope_3d", patch_size=1, hidden_size=1152, depth=28, num_heads=16)
        init = DiT3D(xl, x_shape=(16, 16, 16), max_tokens=5)
    init.init_random(seed=0)  # same seed on every rank: replicas start identical
    tr = DiT3DTrainer(xl, x_shape=(16, 16, 16), max_tokens=5, lr=5e-5, weight_decay=0.01, betas=(0.9, 0.99), max_grad_norm=1.0)
    tr.load_state_dict({k: v.detach() for k, v in init.state_dict().items()}, strict=True)
    del init
    b = args.batch
    g = torch.Generator().manual_seed(100 + rank)
    xs = torch.randn(b, 5, 16, 16, 16, generator=g).cuda()
    noise = torch.randn(b, 10 if diff else 5, 16, 16, 16, generator=g).cuda()
    if diff:  # bash/k600: noise_level random_uniform, variable_context enabled
        tn = dfot_amd.TrainingNoise(noise_level="random_uniform", is_continuous=False, n_context_tokens=2,
                                    variable_context=dfot_amd.ContextTraining(enabled=True, prob=0.25, dropout=0.3))
    else:
        tn = dfot_amd.TrainingNoise(noise_level="random_independent", is_continuous=False, n_context_tokens=2)
    masks = torch.ones(b, 5, dtype=torch.bool)
    levels = [tn.sample(b, 5, masks, g, training=True) for _ in range(args.steps + args.warmup)]

    def train_step(i):
        if diff:
            loss = tr.difference_loss_and_grads(xs, levels[i][0], noise, levels[i][1])
        else:
            loss = tr.loss_and_grads(xs, levels[i][0], noise, levels[i][1])
        tr.optimizer_step(world)
        return loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for i in range(args.warmup):
        train_step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses.append(train_step(args.warmup + i))
    barrier()
    dt = time.perf_counter() - t0
    losses = [float(l.item()) for l in losses]
    assert all(np.isfinite(losses)), losses
    if dist is not None:
        tmax = torch.tensor([dt], device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        n, d, heads = 1280, 72, 16
        video_flop = 28 * (2.0 * n * 1152 * 3456 + 2.0 * n * 1152 * 1152 + 4.0 * n * n * d * heads)
        if diff:  # as bench_k600: 10 merged tokens x 256 patches; spatial block + matrix block per depth
            n, hh, e = 2560, 1152, 64
            lin = lambda rows, k, nn: 2.0 * rows * k * nn
            spatial = lin(n, hh, 3 * hh) + lin(n, hh, hh) + 4.0 * 256 * 256 * 96 * 12 * 10 + 2 * lin(n, hh, 4 * hh)
            matrix = (lin(10 * hh, 256, e) + lin(10 * e, hh, 3 * hh) + 4.0 * 10 * 10 * e * hh + lin(10 * hh, e, 256) + lin(n, hh, hh)
                      + 2 * lin(n, hh, 4 * hh))
            video_flop = 28 * (spatial + matrix)
        step_flop = 3.0 * video_flop * b  # forward + backward (2x), algorithmic
        line = {
            "metric": "training samples/sec, DFoT K600 %s (AdamW, data parallel)" % ("DifferenceDiT3D FacMat XL-64-1 (bash/k600)" if diff else "DiT/XL (per-token independent noise levels)"),
            "value": b * args.steps * world / dt, "unit": "videos/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
            "data": "synthetic latents, seeded random-init weights",
            "config": {"workload": (f"DFoT K600 bash/k600 difference_dit3d factorized_matrix_attention XL-64-1 training step: {b} videos per GPU, latents "
                                    "16x16x16, 5 frames -> 10 merged (difference, frame) tokens, random_uniform levels + variable context, " if diff else
                                    f"DFoT K600 @DiT/XL training step: {b} videos per GPU, latents 16x16x16, 5 tokens, random_independent levels, ") +
                                   "fused_min_snr v-loss, AdamW lr 5e-5 wd 0.01 betas (0.9, 0.99), grad clip 1.0, fp32 master weights / bf16 compute; "
                                   "one all-reduce of the flat fp32 gradient buffer per step", "parameters": tr.numel},
            "losses": losses, "model_tflops": step_flop * args.steps / dt / 1e12,
            "roofline": {"bound": "mfma", "kernel": "whole training step (3 x forward FLOPs: GEMM dgrad/wgrad + attention backward); per-kernel split in profiles/",
                         "achieved": step_flop * args.steps / dt / 1e12, "peak": 2500.0, "unit": "TFLOP/s", "frac": step_flop * args.steps / dt / 1e12 / 2500.0,
                         "traffic": None},
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_train_re10k(args, rank, world, dist):
    """BASELINE config 5: DFoT training step on synthetic RE10K-shaped data -- the UViT3DPose backbone (458.8 M parameters), `--batch` videos of
    8 frames at 256x256 per GPU, per-token independent noise levels (random_independent, continuous), sigmoid-weighted v-prediction loss,
    hand-written backward (uvit_train.UViT3DPoseTrainer: ops over the C ABI), gradient all-reduce, clipped AdamW."""
    import dfot_amd
    from dfot_amd import UViT3DPose
    from dfot_amd.uvit_train import UViT3DPoseTrainer
    init = UViT3DPose(RE10K, x_shape=(3, args.res, args.res), max_tokens=8)
    init.init_random(seed=0)
    cfg = dict(RE10K, resolution=args.res, max_tokens=8, in_channels=3, cond_dim=180, noise_dim=256)
    tr = UViT3DPoseTrainer({k: v.detach() for k, v in init.state_dict().items()}, cfg)
    del init
    b = args.batch
    g = torch.Generator().manual_seed(200 + rank)
    xs = torch.randn(b, 8, 3, args.res, args.res, generator=g).cuda()
    noise = torch.randn(b, 8, 3, args.res, args.res, generator=g).cuda()
    cond = torch.ops.dfot.ray_encoding(synth_poses(b, 8, 300 + rank), args.res)
    tn = dfot_amd.TrainingNoise(noise_level="random_independent", is_continuous=True, n_context_tokens=1)
    masks = torch.ones(b, 8, dtype=torch.bool)
    levels = [tn.sample(b, 8, masks, g, training=True) for _ in range(args.steps + args.warmup)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i):
        # data parallel: the gradient all-reduce overlaps the backward (ranges of the flat buffer reduced in place as levels finish)
        reducer = dfot_amd.parallel.OverlappedGradReducer(flat=tr.flat_grads) if world > 1 else None
        loss = tr.loss_and_grads(xs, cond, levels[i][0], noise, levels[i][1], reducer=reducer)
        tr.optimizer_step(lr=5e-5, betas=(0.9, 0.99), weight_decay=0.01, max_grad_norm=1.0, world_size=world)
        return loss
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    losses = [step(args.warmup + i) for i in range(args.steps)]
    barrier()
    dt = time.perf_counter() - t0
    losses = [float(l.item()) for l in losses]
    assert all(np.isfinite(losses)), losses
    if dist is not None:
        tmax = torch.tensor([dt], device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        step_flop = 3.0 * WINDOW_FLOP * b * (args.res / 256.0) ** 2  # forward + backward (2x), algorithmic
        tf = step_flop * args.steps / dt / 1e12
        line = {
            "metric": "training samples/sec, DFoT RE10K UViT3DPose (per-token independent noise levels, AdamW, data parallel)",
            "value": b * args.steps * world / dt, "unit": "videos/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
            "data": "synthetic frames and camera poses, seeded random-init weights",
            "config": {"workload": f"DFoT RE10K training step (BASELINE config 5): {b} videos x 8 frames x {args.res}x{args.res} per GPU, UViT3DPose "
                                   "(channels 128/256/576/1152, 3+3+6 / 20 blocks), random_independent continuous levels, sigmoid-weighted v-loss, AdamW lr 5e-5 "
                                   "wd 0.01 betas (0.9, 0.99), grad clip 1.0, fp32 master weights / bf16 compute; driver sequencing the C-ABI ops from Python", "parameters": tr.numel},
            "losses": losses, "model_tflops": tf,
            "roofline": {"bound": "mfma", "kernel": "whole training step (3 x forward FLOPs)", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": tf / 2500.0, "traffic": None},
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_k600(args, rank, world, dist):
    """BASELINE config 4: Kinetics-600 latents [16,16,16], 17 frames = 5 latent tokens, context 5 frames = 2 tokens,
    README model @DiT/XL (dit3d full, rope_3d; attention-only blocks in this fork), DiscreteDiffusion cosine / pred_v,
    50 DDIM steps, history guidance 'conditional' (dfot_video.yaml default), `--batch` videos per GPU."""
    import dfot_amd
    from dfot_amd import DFoTVideoSampler, DifferenceDFoTVideoSampler, DifferenceDiT3D, DiffusionConfig, DiT3D, SamplerConfig
    diff = args.workload == "k600diff"  # the bash/k600 model: DifferenceDiT3D factorized matrix attention, XL-64-1, 1358 M parameters
    if diff:
        xl = dict(name="difference_dit3d", variant="factorized_matrix_attention", pos_emb_type="sinusoidal_2d", merge_type="interleaved",
                  patch_size=1, hidden_size=None, embed_col_dim=64, embed_row_dim=1152, num_heads=12, num_col_heads=1, num_row_heads=16,
                  depth=28, mlp_ratio=4.0, spatial_mlp_ratio=4.0, use_bias=True, matrix_block="matrix")
        model = DifferenceDiT3D(xl, x_shape=(16, 16, 16), max_tokens=5).cuda()
    else:
        xl = dict(name="dit3d", variant="full", pos_emb_type="rope_3d", patch_size=1, hidden_size=1152, depth=28, num_heads=16)
        model = DiT3D(xl, x_shape=(16, 16, 16), max_tokens=5).cuda()
    model.init_random(seed=0)
    cfg = SamplerConfig(x_shape=(16, 16, 16), max_tokens=10 if diff else 5,
                        diffusion=DiffusionConfig(sampling_timesteps=args.sampling_steps, beta_schedule="cosine", is_continuous=False),
                        prediction_guidance=dict(name="conditional"))
    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    sampler = (DifferenceDFoTVideoSampler if diff else DFoTVideoSampler)(cfg, model, dfot_amd.device_noise_fn(gen))
    b = args.batch
    xs = torch.randn(b, 5, 16, 16, 16, generator=torch.Generator().manual_seed(rank)).cuda()
    if diff:
        run = lambda: sampler._sample_all_videos(xs, n_context_tokens=2)["prediction"]
    else:
        run = lambda: sampler._predict_videos(xs, n_context_tokens=2, conditions=None)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = (GRAPH_DEFAULT or args.graph) and not args.eager
    sampler.use_graph = use_graph
    for _ in range(args.warmup):
        run()
    if not use_graph and rank == 0:
        model.set_option("time_attn", args.sampling_steps * 28 * args.steps)
    sampler.window_forwards = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    barrier()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    if dist is not None:
        tmax = torch.tensor([dt], device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    fwd = sampler.window_forwards
    roofline_source = "HIP events around every attention launch inside the timed region (eager step loop)"
    if use_graph and rank == 0:  # the same kernels launched eagerly once more, bracketed by events (not part of `value`)
        sampler.use_graph = False
        model.set_option("time_attn", args.sampling_steps * 28)
        run()
        torch.cuda.synchronize()
        sampler.use_graph = True
        roofline_source = "eager pass of one sample right after the timed region, same process (events cannot be timed inside a captured graph)"
    tokens_per_step = 3 * b  # generated latent frames per sample call (5 tokens - 2 context) x videos
    if rank == 0:
        attn_ms, attn_n = model.attn_timing()
        n, d, heads = 1280, 72, 16
        flop_per_launch = 4.0 * n * n * d * heads * b
        # model FLOPs per video-forward: 28 x (qkv 2*N*h*3h + proj 2*N*h*h + attention 4*N^2*d*heads)
        video_flop = 28 * (2.0 * n * 1152 * 3456 + 2.0 * n * 1152 * 1152 + 4.0 * n * n * d * heads)
        if diff:
            # 10 merged tokens x 256 patches; per depth: spatial block (qkv, per-frame attention 12 heads x 96, proj, MLP 4x) +
            # matrix block (left/right factors E=64, frame-token attention, MLP 4x)
            n, h, e = 2560, 1152, 64
            lin = lambda rows, k, nn: 2.0 * rows * k * nn
            spatial = lin(n, h, 3 * h) + lin(n, h, h) + 4.0 * 256 * 256 * 96 * 12 * 10 + 2 * lin(n, h, 4 * h)
            matrix = (lin(10 * h, 256, e) + lin(10 * e, h, 3 * h) + 4.0 * 10 * 10 * e * h + lin(10 * h, e, 256) + lin(n, h, h)
                      + 2 * lin(n, h, 4 * h))
            video_flop = 28 * (spatial + matrix)
            flop_per_launch = 4.0 * 256 * 256 * 96 * 12 * 10 * b  # per-frame spatial attention, 80 x 12 (frame, head) units
        achieved = flop_per_launch * attn_n / (attn_ms * 1e-3) / 1e12 if attn_n else None
        line = {
            "metric": "denoised latent frames/sec, DFoT K600 (%s) 17-frame prediction" % ("DifferenceDiT3D FacMat XL-64-1" if diff else "DiT/XL"), "value": tokens_per_step * args.steps * world / dt,
            "unit": "latent frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
            "data": "synthetic latents, seeded random-init weights",
            "config": {"workload": f"DFoT K600 {'bash/k600 difference_dit3d factorized_matrix_attention XL-64-1 (1358 M params)' if diff else '@DiT/XL'}: {b} videos per GPU, latents 16x16x16, 5 tokens (17 frames), context 2 tokens, "
                                   f"{args.sampling_steps} DDIM steps, conditional history guidance (NFE 1)",
                       "window_forwards_per_step": fwd // args.steps, "frames_per_step": tokens_per_step,
                       "pixel_frames_per_step": 12 * b},
            "video_forward_ms": dt / fwd * 1e3, "model_tflops": video_flop * fwd / dt / 1e12,
            "roofline": {"bound": "mfma", "kernel": ("attn_kernel_v2<128,2,96,96> (per-frame spatial attention, N=256, head dim 96 in 128-wide rows)" if diff else
                                                      "attn_kernel_v2<128,2,80,96> (DiT attention, N=1280, head dim 72 in 128-wide rows)"),
                         "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0 if achieved else None,
                         "traffic": k600_traffic(diff, b), "launches": attn_n, "avg_launch_ms": attn_ms / max(attn_n, 1),
                         "flop_per_launch": flop_per_launch, "source": roofline_source if attn_n else None},
            "sampler_mode": "hipgraph" if use_graph else "eager",
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle import dit as odit
            cores = min(16, os.cpu_count() or 1)
            torch.set_num_threads(cores)
            if diff:
                ocfg = odit.DiffDiTConfig()
                params = odit.diff_seeded_params(ocfg, 0)
                x1, k1, reps, fwd_fn = torch.randn(1, 10, 16, 16, 16), torch.randint(0, 1000, (1, 10)), 1, odit.diff_forward
            else:
                ocfg = odit.DiTConfig()
                params = odit.seeded_params(ocfg, 0)
                x1, k1, reps, fwd_fn = torch.randn(1, 5, 16, 16, 16), torch.randint(0, 1000, (1, 5)), 3, odit.forward
            t1 = time.perf_counter()
            with torch.no_grad():
                for _ in range(reps):
                    fwd_fn(params, ocfg, x1, k1)
            per = (time.perf_counter() - t1) / reps
            line["cpu_baseline"] = {"value": 3.0 / (args.sampling_steps * per), "unit": "latent frames/s", "cores": cores, "kind": "port",
                                    "sample": f"{reps} single-video forward(s) of the oracle {'DifferenceDiT3D' if diff else 'DiT/XL'} ({tuple(x1.shape)}), {per:.2f} s each; scaled to "
                                              f"{args.sampling_steps} forwards per 3 generated latent frames"}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _reduced(j: dict) -> dict:
    """the fields of a child's line that matter under `extra`"""
    keep = {k: j[k] for k in ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "ms_per_step", "dtype", "scaling") if k in j}
    keep["workload"] = j.get("config", {}).get("workload")
    for k in ("model_tflops", "executed_flop_fraction", "window_forward_ms", "video_forward_ms", "sampler_mode", "attention_kernel", "hbm_gbps",
              "ranks_bit_identical", "keyframe_phase", "grad_reduce"):
        if j.get(k) is not None:
            keep[k] = j[k]
    if "roofline" in j:
        keep["roofline"] = {k: j["roofline"].get(k) for k in ("kernel", "achieved", "peak", "frac", "unit", "source")}
    return keep


def extra_jobs(args, world: int) -> dict:
    """BASELINE configs 3, 4 and 5 (and the A/B modes of config 2) as short, labelled runs next to the headline.  One GPU: every
    config.  N > 1 ranks: the two workloads whose multi-GPU form is NOT N independent replicas -- the 200-frame rollout (windows sharded
    over the ranks, key-frame History-Guidance branches split over rank pairs, one all-gather per plan stage) and the RE10K training
    step (gradient all-reduce overlapped with the backward) -- so that the driver's `--gpus N` run measures them on RCCL."""
    if world > 1 or args.dry_run:
        jobs = {"200f": ["--workload", "200f", "--steps", "1", "--warmup", "0"]}
        if not args.dry_run:
            jobs["train_re10k"] = ["--workload", "train_re10k", "--batch", "8", "--steps", "2", "--warmup", "1"]
        return jobs
    return {
        "200f": ["--workload", "200f", "--steps", "1", "--warmup", "0"],                       # config 3, full 50 DDIM steps
        "k600": ["--workload", "k600", "--steps", "2", "--warmup", "1"],                       # config 4, README @DiT/XL
        "k600diff": ["--workload", "k600diff", "--steps", "1", "--warmup", "1"],               # config 4, bash/k600 model
        "train_re10k": ["--workload", "train_re10k", "--batch", "8", "--steps", "2", "--warmup", "1"],  # config 5
        "train_k600": ["--workload", "train_k600", "--steps", "3", "--warmup", "1"],
        ("8f_eager" if GRAPH_DEFAULT else "8f_graph"): ["--workload", "8f", "--steps", "2", "--warmup", "1"] + (["--eager"] if GRAPH_DEFAULT else ["--graph"]),
        # what a checkpoint whose q_norm / k_norm weights exceed the no-running-max bound would get: level-2 attention on attention_v3 (variant 5)
        "8f_safe_attention": ["--workload", "8f", "--steps", "2", "--warmup", "1", "--attn-safe"],
        "vae_decode": ["--workload", "vae_decode", "--steps", "3", "--warmup", "1"],
    }


def coordinate(args):
    """The process the driver starts (one per rank under `torch.distributed.run`, or the single `python bench.py`).  It NEVER touches
    the GPU: the headline and every extra run as fresh `--worker` children, one after the other, so each starts on a clean device and a
    failing or hanging extra cannot cost the headline line.  With N > 1 ranks the N coordinators agree on one rendezvous port per job
    over a CPU (gloo) group and keep in step with barriers; rank 0 assembles and prints the ONE JSON line."""
    import datetime
    import socket
    import subprocess
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    jobs = {} if args.no_extras else extra_jobs(args, world)
    cdist = None
    ports = [0] * (1 + len(jobs))
    if world > 1:
        import torch.distributed as cdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        cdist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=40))
        if rank == 0:
            socks = [socket.socket() for _ in ports]
            for sk in socks:
                sk.bind(("127.0.0.1", 0))
            ports = [sk.getsockname()[1] for sk in socks]
            for sk in socks:
                sk.close()
        box = [ports]
        cdist.broadcast_object_list(box, src=0)
        ports = box[0]
    base = [sys.executable, os.path.abspath(__file__), "--worker", "--sampling-steps", str(args.sampling_steps), "--res", str(args.res)] + \
           (["--dry-run"] if args.dry_run else [])
    # a child is a rank of ITS OWN job: same RANK / WORLD_SIZE, a fresh port, nothing of the launcher's agent-store settings
    env0 = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_") and k not in ("GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE")}

    def run(job_index: int, job_args, timeout: float):
        env = dict(env0, RANK=str(rank), LOCAL_RANK=str(local), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(ports[job_index]))
        t0 = time.perf_counter()
        try:
            r = subprocess.run(base + job_args, capture_output=True, text=True, timeout=timeout, env=env)
            rc, out, err = r.returncode, r.stdout, r.stderr
        except subprocess.TimeoutExpired as e:
            rc, out, err = -9, "", f"timeout after {timeout:.0f} s: {e}"
        if cdist is not None:  # every rank's child has ended (or was killed) before the next job's rendezvous starts
            ok = [None] * world
            cdist.all_gather_object(ok, rc)
            rc = next((c for c in ok if c), 0)
        rows = [l for l in out.splitlines() if l.startswith("{")]
        return rc, (json.loads(rows[-1]) if rows else None), err, time.perf_counter() - t0

    head_args = ["--workload", "8f", "--steps", str(args.steps), "--warmup", str(args.warmup)] + (["--no-cpu-baseline"] if args.no_cpu_baseline else []) + \
                (["--eager"] if args.eager else []) + (["--graph"] if args.graph else []) + (["--attn-safe"] if args.attn_safe else [])
    rc, line, err, _ = run(0, head_args, 1500)
    if rc != 0 or (rank == 0 and line is None):
        sys.stderr.write(err[-4000:])
        raise SystemExit(rc if rc > 0 else 1)
    extra = {}
    for i, (name, job_args) in enumerate(jobs.items()):
        try:
            rc, j, err, wall = run(1 + i, ["--no-cpu-baseline"] + job_args, 900)
            if rc != 0 or (rank == 0 and j is None):
                extra[name] = {"error": (err or "")[-300:], "returncode": rc}
                continue
            if rank == 0:
                extra[name] = dict(_reduced(j), wall_s=wall)
        except Exception as e:  # an extra must never cost the headline line
            extra[name] = {"error": repr(e)[:300]}
    if world > 1 and args.dry_run and not args.no_extras:
        extra["train_re10k"] = {"skipped": "dry run: the training step has no host-only form"}
    if rank == 0:
        if jobs or extra:
            line["extra"] = extra
        print(json.dumps(line), flush=True)
    if cdist is not None:
        cdist.barrier()
        cdist.destroy_process_group()


def bench_vae_decode(args):
    """K600 latents -> frames, the step after the sampler for latent datasets (base_pytorch_video_algo.py:553-629 `_decode`):
    VideoVAE decoder (default widths 128 x (1, 2, 4, 4), 2 ResBlocks per level, z 16), `--batch` videos of 5 latent tokens
    (16 x 16) -> 17 frames of 128 x 128.  Reports ms per video and the algorithmic HBM bytes of the GroupNorm / upsample passes."""
    import dfot_amd
    dec = dfot_amd.VideoVAEDecoder(z_channels=16, hidden_size=128, embed_dim=16).cuda()
    dec.init_random(seed=0)
    b = min(args.batch, 2)  # vae.batch_size of the K600 configs
    z = torch.randn(b, 5, 16, 16, 16, generator=torch.Generator().manual_seed(0)).cuda()
    run = lambda: dfot_amd.decode_latents(dec, z, n_frames=17, vae_batch_size=b)
    for _ in range(args.warmup):
        out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    assert torch.isfinite(out).all() and tuple(out.shape) == (b, 17, 3, 128, 128)
    line = {"metric": "VideoVAE decode, K600 latents -> 17 frames of 128x128", "value": b * 17 / dt, "unit": "pixel frames/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic latents, seeded random-init weights",
            "config": {"workload": f"VideoVAE decoder: {b} videos x 5 latent tokens (16x16x16) -> 17 frames 128x128 per call", "ms_per_video": dt * 1e3 / b}}
    print(json.dumps(line), flush=True)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: this parent starts N fresh rank processes, one per GPU, with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- what `accelerate launch --multi_gpu --num_processes N` does for the reference
    (configurations/cluster/a2i2_multigpu.yaml:45-57) -- forwards rank 0's JSON line and exits non-zero if any rank does.  A rank that
    dies takes the others down instead of leaving them in a collective.  The environment is inherited as it is."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # its own session: the rank and the workers it starts form one process group that can be ended together
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, start_new_session=True,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    failed = None
    while failed is None and any(p.poll() is None for p in procs[1:]) and procs[0].poll() is None:
        time.sleep(0.2)
        failed = next((p for p in procs if p.poll() not in (None, 0)), None)
    if failed is not None:
        import signal
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
    out0 = procs[0].stdout.read() if procs[0].stdout else ""
    rcs = [p.wait() for p in procs]
    for l in out0.splitlines():
        if l.startswith("{"):
            print(l, flush=True)
    if any(rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs}\n")
        raise SystemExit(next(rc for rc in rcs if rc) if any(rc > 0 for rc in rcs) else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--sampling-steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="8f only: skip the short runs of the other BASELINE configs appended under 'extra'")
    ap.add_argument("--worker", action="store_true", help="internal: this process IS a rank of one workload (started by the coordinator)")
    ap.add_argument("--graph", action="store_true", help="hipGraph step loop (the default): all remaining DDIM steps of a window as ONE graph")
    ap.add_argument("--eager", action="store_true", help="force the eager step loop (A/B against the default)")
    ap.add_argument("--attn-safe", action="store_true", help="level-2 attention on the running-max kernel (what weights beyond the QK-norm bound get)")
    ap.add_argument("--dry-run", action="store_true", help="host plumbing only (CPU, gloo): plans, shards and gathers every window but launches nothing")
    ap.add_argument("--batch", type=int, default=8, help="k600: videos per GPU (bash/k600 validation.batch_size)")
    ap.add_argument("--workload", choices=["8f", "200f", "k600", "k600diff", "train_k600", "train_k600diff", "train_re10k", "vae_decode"], default="8f",
                    help="8f: BASELINE config 2 (default, the metric's single-GPU configuration); 200f: config 3, the "
                         "200-frame rollout (keyframe density 0.0625, stabilized HG 4.0/0.02 + interpolation HG 1.5, batches of "
                         "4 windows), interpolation windows sharded over ranks")
    args = ap.parse_args()
    use_graph = (GRAPH_DEFAULT or args.graph) and not args.eager

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not args.worker and args.workload == "8f" and not args.no_extras:
        # the driver's command: headline (config 2) + the other BASELINE configs, each a fresh worker process of its own
        return coordinate(args)
    dev = "cpu" if args.dry_run else "cuda"
    if not args.dry_run:
        torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import dfot_amd
    from dfot_amd import DFoTVideoPoseSampler, DiffusionConfig, SamplerConfig, UViT3DPose

    if args.dry_run and args.workload not in ("8f", "200f"):
        raise SystemExit("--dry-run covers the sampler workloads (8f, 200f)")
    if args.workload == "vae_decode":
        return bench_vae_decode(args)
    if args.workload == "train_re10k":
        return bench_train_re10k(args, rank, world, dist if world > 1 else None)
    if args.workload in ("train_k600", "train_k600diff"):
        return bench_train_k600(args, rank, world, dist if world > 1 else None)
    if args.workload in ("k600", "k600diff"):
        return bench_k600(args, rank, world, dist if world > 1 else None)
    res = args.res
    model = None
    if not args.dry_run:
        model = UViT3DPose(RE10K, x_shape=(3, res, res), max_tokens=8).cuda()
        model.init_random(seed=0)
        if args.attn_safe:
            model.set_option("attn_force_safe", 1)
    long_rollout = args.workload == "200f"
    if long_rollout:
        n_frames = 200
        cfg = SamplerConfig(x_shape=(3, res, res), max_tokens=8,
                            diffusion=DiffusionConfig(sampling_timesteps=args.sampling_steps),
                            prediction_guidance=dict(name="stabilized_vanilla", guidance_scale=4.0, stabilization_level=0.02),
                            interpolation_guidance=dict(name="vanilla", guidance_scale=1.5), keyframe_density=0.0625,
                            interpolation_max_batch_size=4)
        # every rank works on the SAME video; noise is keyed by window so replicated key-frame windows agree bitwise
        sampler = DFoTVideoPoseSampler(cfg, model, dfot_amd.parallel.WindowKeyedNoise(1234, device=dev))
        sampler.shard_windows = world > 1
        # the sequential key-frame windows: History-Guidance branches split over rank pairs (one all-gather of v per step)
        sampler.branch_parallel = world > 1
        xs = torch.randn(1, n_frames, 3, res, res, generator=torch.Generator().manual_seed(0)).to(dev)
        conds = synth_poses(1, n_frames, 100).to(dev)
    else:
        n_frames = 8
        cfg = SamplerConfig(x_shape=(3, res, res), max_tokens=8,
                            diffusion=DiffusionConfig(sampling_timesteps=args.sampling_steps),
                            prediction_guidance=dict(name="vanilla", guidance_scale=4.0))
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        sampler = DFoTVideoPoseSampler(cfg, model, dfot_amd.device_noise_fn(gen) if not args.dry_run else
                                       (lambda tag, shape: torch.randn(shape, generator=gen)))
        xs = torch.randn(1, n_frames, 3, res, res, generator=torch.Generator().manual_seed(rank)).to(dev)
        conds = synth_poses(1, n_frames, 100 + rank).to(dev)
    if args.dry_run:
        sampler.device, sampler.dry_run = "cpu", True

    def barrier():
        if world > 1:
            dist.barrier()
        if not args.dry_run:
            torch.cuda.synchronize()

    sampler.use_graph = use_graph and not args.dry_run
    for _ in range(args.warmup):
        sampler._predict_videos(xs, n_context_tokens=1, conditions=conds)
    # HIP events around the level-2 attention launches need the eager loop (hipEventElapsedTime refuses events recorded as nodes of a
    # captured graph, "invalid resource handle" on ROCm 7.2): in hipGraph mode the roofline comes from ONE labelled eager sample run
    # right after the timed region, in this process; in eager mode every launch inside the timed region is event-timed
    timed_attn = not sampler.use_graph and not args.dry_run
    attn_per_sample = args.sampling_steps * 12 * (14 if long_rollout else 1)
    if model is not None and timed_attn:
        model.set_option("time_attn", attn_per_sample * args.steps if rank == 0 else 0)
    sampler.window_forwards = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = sampler._predict_videos(xs, n_context_tokens=1, conditions=conds)
    barrier()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    ranks_identical = None
    if world > 1:
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        if long_rollout:  # all ranks cooperate on ONE video: every rank must end with the same 200 frames, bit for bit
            ref = out.clone()
            dist.broadcast(ref, 0)
            same = torch.tensor([1.0 if torch.equal(ref, out) else 0.0], device=dev)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            ranks_identical = bool(same.item() == 1.0)
            assert ranks_identical, "200f: the ranks ended with different rollouts"
    frames_per_sample = n_frames - 1
    # 8f: one video per rank (weak scaling); 200f: all ranks cooperate on one video (strong scaling)
    total_frames = frames_per_sample * args.steps * (1 if long_rollout else world)
    fwd = sampler.window_forwards
    roofline_source = "HIP events around every level-2 attention launch inside the timed region (eager step loop)"
    roofline_fwd = fwd
    if model is not None and sampler.use_graph and rank == 0 and not long_rollout:
        # the same kernels, launched eagerly once more so that they can be bracketed by events (not part of `value`)
        sampler.use_graph = False
        model.set_option("time_attn", attn_per_sample)
        sampler.window_forwards = 0
        sampler._predict_videos(xs, n_context_tokens=1, conditions=conds)
        torch.cuda.synchronize()
        roofline_fwd = sampler.window_forwards
        sampler.use_graph = True
        roofline_source = "eager pass of one sample right after the timed region, same process (events cannot be timed inside a captured graph)"

    if rank == 0:
        attn_ms, attn_n = model.attn_timing() if model is not None else (0.0, 0)
        n2 = 8 * (res // 8) ** 2
        # total level-2 attention FLOPs of the event-timed launches = 4*N^2*d*heads per (window-forward, block) x 12 blocks
        # x window-forwards; achieved = FLOPs / summed duration.  flop_per_launch is quoted for the model-batch-2 launch of 8f.
        flop_per_launch = 4.0 * n2 * n2 * 64 * 9 * 2
        total_attn_flop = 4.0 * n2 * n2 * 64 * 9 * 12 * roofline_fwd
        achieved = total_attn_flop / (attn_ms * 1e-3) / 1e12 if attn_n else None
        traffic = traffic_source = None
        pmc = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r04_g_pmc_hbm_traffic_8f.json", "r03_pmc_hbm_traffic_8f.json")) if os.path.exists(f)), None)
        if pmc and res == 256 and not long_rollout:
            ks = [v for n, v in json.load(open(pmc))["kernels"].items() if "attn64_kernel_v5" in n]
            traffic = ks[0]["hbm_bytes_per_launch_corrected"] if ks else None
            traffic_source = "committed PMC pass of the same command (%s), not measured in this run" % os.path.relpath(pmc, ROOT)
        akern = int(model.query("attn_kernel_l2")) if model is not None else None
        kname = {14: "attn64_kernel_v5 (no running max: QK-norm bound < 64)", 5: "attn64_kernel_v3 (running max)"}.get(akern, "none (dry run)")
        # rows of the model batch whose frame-local work is skipped: 8f has one context frame of eight -- dead in both branches (1/8 of
        # the rows), frozen in the conditional branch from a window's second step on (1/16 of the rows)
        exec_frac = None if long_rollout or res % 8 else executed_flop_fraction(res, 8, 1 / 8, (1 / 16) * (1 - 1 / args.sampling_steps) if (res // 8) ** 2 % 512 == 0 else 0.0)
        nominal = WINDOW_FLOP * (res / 256.0) ** 2 * fwd / dt / 1e12
        line = {
            "metric": "denoised latent frames/sec, DFoT RE10K 8f & 200f rollout @1/2/4/8 GPU",
            "value": total_frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if long_rollout else "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic latents, seeded random-init weights",
            "config": {"workload": (f"DFoT_RE10K 200-frame rollout: 12 key frames (stabilized HG 4.0/0.02) + 2-stage interpolation "
                                    f"(HG 1.5, batches of 4 windows), {args.sampling_steps} DDIM steps, {res}x{res}, windows sharded "
                                    f"over ranks, key-frame HG branches split over rank pairs") if long_rollout else
                                   (f"DFoT_RE10K 8-frame sample: context 1, {args.sampling_steps} DDIM steps, vanilla history "
                                    f"guidance 4.0 (NFE 2), {res}x{res}, one video per GPU"),
                       "window_forwards_per_step": fwd // args.steps, "frames_per_step": frames_per_sample},
            "rccl_ranks": dist.get_world_size() if world > 1 else 1, "collective_backend": dist.get_backend() if world > 1 else None,
            "window_forward_ms": dt / fwd * 1e3,
            # NOMINAL rate: the reference's FLOPs per window-forward over the measured time.  The engine executes
            # only `executed_flop_fraction` of them (per-window FiLM cache, dead / frozen frame skips: DESIGN.md 4a), so the hardware
            # utilisation is model_tflops_executed / 2500, not model_tflops / 2500
            "model_tflops": nominal, "executed_flop_fraction": exec_frac,
            "model_tflops_executed": nominal * exec_frac if exec_frac else None,
            "roofline": {"bound": "mfma", "kernel": "%s + attn64_merge_kernel (level-2 flash attention, N=%d, d=64; the event pair "
                                                   "brackets both launches)" % (kname.split(" (")[0], n2),
                         "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": achieved / 2500.0 if achieved else None, "traffic": traffic, "traffic_source": traffic_source,
                         "launches": attn_n, "avg_launch_ms": attn_ms / max(attn_n, 1),
                         "flop_per_launch": flop_per_launch, "source": roofline_source if attn_n else None},
            # which level-2 attention kernel these weights got and why: the fast kernel needs the bound below 64
            "attention_kernel": kname, "score_bound": model.query("score_bound_l2") if model is not None else None,
        }
        if long_rollout and world > 1:
            line["keyframe_phase"] = (f"{world // 2} rank pair(s) each evaluate the two History-Guidance branches of the sequential key-frame windows; "
                                      "pairs beyond the first repeat the first pair's work (replicated state, no broadcast) -- that phase does not "
                                      "scale past 2 ranks" + ("" if world % 2 == 0 else "; odd world: no branch split, every rank evaluates both branches"))
        if not args.no_cpu_baseline and world == 1 and not args.dry_run:  # reported baseline: rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(res, 1, frames_per_sample, fwd // args.steps)
        line["sampler_mode"] = "dry-run (host plumbing only, nothing launched)" if args.dry_run else ("hipgraph" if sampler.use_graph else "eager")
        if sampler.use_graph and not args.dry_run:  # captures paid (once per window shape) and steps replayed, warm-up included
            line["graph_captures"], line["graph_step_replays"] = sampler.graph_captures, sampler.graph_replays
        if ranks_identical is not None:
            line["ranks_bit_identical"] = ranks_identical
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
