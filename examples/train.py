#!/usr/bin/env python3
"""Training loop with the MI355X engine, the way the reference's Lightning trainer drives `training_step`.

  python examples/train.py k600     [--ckpt K600.ckpt] [--steps 100] [--batch 8] [--save out.ckpt]
  python examples/train.py k600diff [--accumulate 2]                                  # the model bash/k600/*.sh train
  python examples/train.py re10k    [--batch 8]                                       # RE10K UViT3DPose (BASELINE config 5), synthetic frames + poses
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/train.py k600   # data parallel, one rank per GPU

Data are synthetic latents (no dataset offline); everything else is the reference's recipe: per-token noise levels from
`_get_training_noise_levels` (random_independent for @DiT/XL, random_uniform + variable context for bash/k600), fused-min-SNR
v-loss, AdamW lr 5e-5 / wd 0.01 / betas (0.9, 0.99), gradient clipping 1.0, gradients averaged over the ranks.
The saved file uses the reference's key names (`diffusion_model.model.*`) and loads into the reference or into the samplers here.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dfot_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", choices=["k600", "k600diff", "re10k"])
    ap.add_argument("--ckpt")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--accumulate", type=int, default=1)
    ap.add_argument("--lr", type=float, default=5e-5)
    ap.add_argument("--warmup-steps", type=int, default=10000, help="re10k: linear lr warm-up (constant_with_warmup, realestate10k_video_generation.yaml)")
    ap.add_argument("--save")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    if a.model == "re10k":
        return train_re10k(a, rank, world)
    diff = a.model == "k600diff"
    if diff:
        cfg = dict(name="difference_dit3d", variant="factorized_matrix_attention", pos_emb_type="sinusoidal_2d", merge_type="interleaved",
                   patch_size=1, embed_col_dim=64, embed_row_dim=1152, num_heads=12, num_col_heads=1, num_row_heads=16, depth=28,
                   mlp_ratio=4.0, spatial_mlp_ratio=4.0, use_bias=True, matrix_block="matrix")
        init = dfot_amd.DifferenceDiT3D(cfg, x_shape=(16, 16, 16), max_tokens=5)
        sampling = dfot_amd.TrainingNoise(noise_level="random_uniform", is_continuous=False, n_context_tokens=2,
                                          variable_context=dfot_amd.ContextTraining(enabled=True, prob=0.25, dropout=0.3))
    else:
        cfg = dict(name="dit3d", variant="full", pos_emb_type="rope_3d", patch_size=1, hidden_size=1152, depth=28, num_heads=16)
        init = dfot_amd.DiT3D(cfg, x_shape=(16, 16, 16), max_tokens=5)
        sampling = dfot_amd.TrainingNoise(noise_level="random_independent", is_continuous=False, n_context_tokens=2)
    trainer = dfot_amd.DiT3DTrainer(cfg, x_shape=(16, 16, 16), max_tokens=5, lr=a.lr)
    if a.ckpt:
        dfot_amd.load_reference_checkpoint(trainer, a.ckpt)
    else:
        init.init_random(seed=0)  # the same on every rank
        trainer.load_state_dict({k: v.detach() for k, v in init.state_dict().items()})
    del init
    g = torch.Generator().manual_seed(1000 + rank)
    masks = torch.ones(a.batch, 5, dtype=torch.bool)
    t0 = time.perf_counter()
    for step in range(a.steps):
        for _ in range(a.accumulate):
            frames = torch.randn(a.batch, 5, 16, 16, 16, generator=g)
            noise = torch.randn(a.batch, 10 if diff else 5, 16, 16, 16, generator=g)
            levels, loss_masks = sampling.sample(a.batch, 5, masks, g, training=True)
            loss = (trainer.difference_loss_and_grads if diff else trainer.loss_and_grads)(frames, levels, noise, loss_masks)
            if a.accumulate > 1:
                trainer.accumulate()
        trainer.optimizer_step(world)
        if rank == 0 and (step % 5 == 0 or step == a.steps - 1):
            print(f"step {step:4d}  loss {float(loss.item()):.4f}  {(time.perf_counter() - t0) / (step + 1) * 1e3:.1f} ms/step", flush=True)
    if a.save and rank == 0:
        torch.save({"state_dict": {"diffusion_model.model." + k: v.cpu() for k, v in trainer.state_dict().items()}}, a.save)
        print("saved", a.save)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def train_re10k(a, rank, world):
    """DFoTVideoPose training on the op-by-op UViT3DPose driver: per-token independent continuous levels, sigmoid-weighted v-loss"""
    from bench import RE10K, synth_poses
    from dfot_amd import parallel
    from dfot_amd.training import lr_at_step
    from dfot_amd.uvit_train import UViT3DPoseTrainer
    dcfg = dfot_amd.DiffusionConfig()  # training schedule (cosine, shift 0.125), sigmoid loss weighting (bias -1), precond_scale 0.125
    init = dfot_amd.UViT3DPose(RE10K, x_shape=(3, 256, 256), max_tokens=8)
    if a.ckpt:
        dfot_amd.load_reference_checkpoint(init, a.ckpt)
    else:
        init.init_random(seed=0)
    trainer = UViT3DPoseTrainer({k: v.detach() for k, v in init.state_dict().items()}, dict(RE10K, resolution=256, max_tokens=8))
    del init
    sampling = dfot_amd.TrainingNoise(noise_level="random_independent", is_continuous=True, n_context_tokens=1)
    g = torch.Generator().manual_seed(1000 + rank)
    masks = torch.ones(a.batch, 8, dtype=torch.bool)
    t0 = time.perf_counter()
    for step in range(a.steps):
        frames = torch.randn(a.batch, 8, 3, 256, 256, generator=g)
        noise = torch.randn(a.batch, 8, 3, 256, 256, generator=g)
        cond = torch.ops.dfot.ray_encoding(synth_poses(a.batch, 8, 7 * step + rank), 256)
        levels, loss_masks = sampling.sample(a.batch, 8, masks, g, training=True)
        reducer = parallel.OverlappedGradReducer() if world > 1 else None   # gradient all-reduce overlapped with the backward
        loss = trainer.loss_and_grads(frames, cond, levels, noise, loss_masks, diffusion=dcfg, reducer=reducer)
        lr = lr_at_step(step, a.lr, "constant_with_warmup", a.warmup_steps)  # realestate10k_video_generation.yaml:19-22
        trainer.optimizer_step(lr=lr, world_size=world)
        if rank == 0 and (step % 5 == 0 or step == a.steps - 1):
            print(f"step {step:4d}  loss {float(loss.item()):.4f}  {(time.perf_counter() - t0) / (step + 1) * 1e3:.1f} ms/step", flush=True)
    if a.save and rank == 0:
        torch.save({"state_dict": {"diffusion_model.model." + k: v.cpu() for k, v in trainer.state_dict().items()}}, a.save)
        print("saved", a.save)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
