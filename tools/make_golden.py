#!/usr/bin/env python3
"""Generate tests/golden/*.npz by executing the reference's own source on CPU.

Run ONLY in the build container (needs /root/reference):   python tools/make_golden.py
The output fixtures are data (inputs, injected noise, expected outputs) -- no reference source.
Weights are never stored: they come from oracle.uvit.seeded_params(cfg, seed) which this
script loads into the reference module (strict) and the tests re-create bit-identically; a
checksum of the weights is stored with each fixture.

Fixtures
  schedule.npz        DiscreteDiffusion buffers for RE10K (cosine_simple_diffusion, shifted 0.125),
                      ddim level table, scheduling matrices
  schedule_extra.npz  interleaved / gibbs / autoregressive scheduling matrices (7 sampling steps)
  ray_encoding.npz    DFoTVideoPose._process_conditions at resolution 8 and sampled rows at 256
  backbone_w64.npz    UViT3DPose.forward, RE10K widths [128,256,576,1152], resolution 64, Bm=2
  backbone_tiny.npz   UViT3DPose.forward, tiny widths, resolution 16
  step_trace.npz      one sample_step + HG prepare/compose per scheme (tiny model)
  sampler_8f.npz      _predict_videos 8 frames / context 1 / 3 DDIM steps / vanilla(4.0) (tiny model)
  sampler_200f.npz    _predict_videos 200 frames, keyframe density 0.0625, stabilized_vanilla(4,0.02)
                      + interpolation vanilla(1.5) max_batch 4, 2 DDIM steps (tiny model) + plan trace
  training_loss.npz   ContinuousDiffusion.forward loss on the tiny model with injected noise
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ref_loader  # noqa: E402
from oracle import pose as opose  # noqa: E402
from oracle import uvit as ouvit  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def weights_digest(params) -> str:
    h = hashlib.sha256()
    for k in params:
        h.update(k.encode())
        h.update(params[k].contiguous().numpy().tobytes())
    return h.hexdigest()


TINY = dict(channels=[32, 64, 72, 144], emb_channels=64, num_updown_blocks=[1, 1, 2], num_mid_blocks=2,
            num_heads=9)
W64 = dict(channels=[128, 256, 576, 1152], emb_channels=1024, num_updown_blocks=[3, 3, 6], num_mid_blocks=20,
           num_heads=9)


def algo_cfg(A, res: int, backbone: dict, n_frames: int = 8, context: int = 1, sampling_steps: int = 50,
             pred_hg=None, interp_hg=None, density=None, interp_enabled=False, max_batch=None):
    hgc = lambda d: dict(d, visualize=False)
    return A(dict(
        debug=False, lr=5e-5, x_shape=[3, res, res], max_frames=8, n_frames=n_frames, frame_skip=1,
        context_frames=context,
        latent=dict(enabled=False, type="pre_sample", suffix=None, downsampling_factor=[1, 1], shape=None,
                    num_channels=3),
        data_mean=[[[0.577]], [[0.517]], [[0.461]]], data_std=[[[0.249]], [[0.249]], [[0.268]]],
        external_cond_type="action", external_cond_num_classes=None, external_cond_dim=16,
        external_cond_stack=False, external_cond_processing=None,
        compile=False, weight_decay=0.01, optimizer_beta=[0.9, 0.99],
        lr_scheduler=dict(name="constant_with_warmup", num_warmup_steps=10),
        noise_level="random_independent", uniform_future=dict(enabled=False),
        fixed_context=dict(enabled=False, indices=None, dropout=0),
        variable_context=dict(enabled=False, prob=0.25, dropout=0.3),
        chunk_size=-1, scheduling_matrix="full_sequence", replacement="noisy_scale",
        refinement_sampling=dict(enabled=False, goback_length=20, n_goback=5),
        save_attn_map=dict(enabled=False, attn_map_dir=None),
        diffusion=dict(
            is_continuous=True, precond_scale=0.125, timesteps=1000, beta_schedule="cosine_simple_diffusion",
            schedule_fn_kwargs=dict(shifted=0.125, interpolated=False), use_causal_mask=False, clip_noise=20.0,
            objective="pred_v", loss_weighting=dict(strategy="sigmoid", sigmoid_bias=-1.0),
            training_schedule=dict(name="cosine", shift=0.125), sampling_timesteps=sampling_steps,
            ddim_sampling_eta=0.0, reconstruction_guidance=0.0),
        vae=dict(pretrained_path=None, pretrained_kwargs={}, use_fp16=False, batch_size=2),
        checkpoint=dict(reset_optimizer=False, strict=True),
        tasks=dict(
            prediction=dict(enabled=True, history_guidance=hgc(pred_hg or dict(name="conditional")),
                            keyframe_density=density, sliding_context_len=None),
            interpolation=dict(enabled=interp_enabled, history_guidance=hgc(interp_hg or dict(name="conditional")),
                               max_batch_size=max_batch)),
        logging=dict(deterministic=0, loss_freq=100, grad_norm_freq=100, max_num_videos=8, n_metrics_frames=None,
                     metrics=[], metrics_batch_size=16, sanity_generation=False, raw_dir=None),
        camera_pose_conditioning=dict(normalize_by="first", bound=None, type="ray_encoding"),
        backbone=dict(
            name="u_vit3d_pose", patch_size=2,
            block_types=["ResBlock", "ResBlock", "TransformerBlock", "TransformerBlock"],
            block_dropouts=[0.0, 0.0, 0.1, 0.1], pos_emb_type="rope", use_checkpointing=[False] * 4,
            conditioning=dict(dim=None), external_cond_dropout=0.1, use_fourier_noise_embedding=True,
            **backbone),
    ))


def build_algo(R, cfg, seed=0, res=None):
    algo = R["DFoTVideoPose"](cfg)
    algo.eval()
    ocfg = oracle_cfg(cfg)
    params = ouvit.seeded_params(ocfg, seed)
    model = algo.diffusion_model.model
    missing, unexpected = model.load_state_dict(params, strict=False)
    persistent_missing = [k for k in missing if not k.startswith("pos_embs.")]
    assert not persistent_missing and not unexpected, (persistent_missing, unexpected)
    return algo, ocfg, params


def oracle_cfg(cfg) -> ouvit.UViTConfig:
    b = cfg.backbone
    return ouvit.UViTConfig(channels=tuple(b.channels), emb_channels=b.emb_channels,
                            num_updown_blocks=tuple(b.num_updown_blocks), num_mid_blocks=b.num_mid_blocks,
                            num_heads=b.num_heads, resolution=cfg.x_shape[-1], max_tokens=8)


def synth_poses(b: int, t: int, seed: int = 0) -> torch.Tensor:
    """Synthetic RE10K-format raw poses (B,T,16): intrinsics (0.5,0.9,0.5,0.5)+jitter, small random
    rotations about all axes, translation drifting along x/z."""
    g = torch.Generator().manual_seed(1000 + seed)
    k = torch.tensor([0.5, 0.9, 0.5, 0.5]).repeat(b, t, 1) + 0.02 * torch.randn(b, 1, 4, generator=g)
    ang = 0.15 * torch.randn(b, t, 3, generator=g)
    cx, sx, cy, sy, cz, sz = ang[..., 0].cos(), ang[..., 0].sin(), ang[..., 1].cos(), ang[..., 1].sin(), \
        ang[..., 2].cos(), ang[..., 2].sin()
    one, zero = torch.ones_like(cx), torch.zeros_like(cx)
    rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], -1).view(b, t, 3, 3)
    ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], -1).view(b, t, 3, 3)
    rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], -1).view(b, t, 3, 3)
    rot = rz @ ry @ rx
    trans = torch.stack([torch.linspace(0, 0.5, t).repeat(b, 1), 0.05 * torch.randn(b, t, generator=g),
                         torch.linspace(0, -0.3, t).repeat(b, 1)], -1) + 0.3 * torch.randn(b, 1, 3, generator=g)
    rt = torch.cat([rot, trans[..., None]], -1).reshape(b, t, 12)
    return torch.cat([k, rt], -1).float()


class RandnRecorder:
    """Records every normal draw the reference makes (torch.randn / torch.randn_like)."""

    def __init__(self):
        self.draws = []
        self._randn, self._randn_like = torch.randn, torch.randn_like

    def __enter__(self):
        rec = self

        def randn(*a, **k):
            t = rec._randn(*a, **k)
            rec.draws.append(t.clone())
            return t

        def randn_like(x, **k):
            t = rec._randn_like(x, **k)
            rec.draws.append(t.clone())
            return t
        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *a):
        torch.randn, torch.randn_like = self._randn, self._randn_like


def save(name, **arrays):
    path = os.path.join(OUT, name)
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = v
    np.savez_compressed(path, **conv)
    print(f"  wrote {name}: {os.path.getsize(path) / 1e6:.2f} MB")


TRAIN = dict(channels=[128, 128, 128, 256], emb_channels=128, num_updown_blocks=[1, 1, 1], num_mid_blocks=1, num_heads=2)


@torch.enable_grad()
def training_grads_uvit(R):
    """training_grads_uvit.npz: the reference's own training loss for the pose model -- ContinuousDiffusion.forward (per-token levels in [0,1],
    v-prediction, sigmoid weighting) through UViT3DPose (reduced widths, 128x128 frames, 8 tokens, dropout off) -> _reweight_loss with
    masks -> backward(): loss, the L2 norm of every parameter gradient and the full gradient of the small tensors"""
    print("training grads (UViT3DPose)")
    A = R["AttrDict"]
    cfg = algo_cfg(A, 128, TRAIN)
    algo, ocfg, params = build_algo(R, cfg, seed=6)
    g = torch.Generator().manual_seed(31)
    xs = torch.randn(1, 8, 3, 128, 128, generator=g)
    k = torch.rand(1, 8, generator=g)
    masks = torch.ones(1, 8)
    masks[0, 5] = 0
    poses = synth_poses(1, 8, seed=2)
    model = algo.diffusion_model.model
    for p_ in model.parameters():
        p_.grad = None
    cond = algo._process_conditions(poses.clone())
    with RandnRecorder() as rec:
        _, loss = algo.diffusion_model(xs, cond, k=k)
    loss = algo._reweight_loss(loss, masks)
    loss.backward()
    grads = {n: p_.grad.detach().clone() for n, p_ in model.named_parameters() if p_.grad is not None}
    out = {"loss": loss.detach(), "noise": rec.draws[0], "names": np.array(list(grads)),
           "norms": np.array([float(v.norm()) for v in grads.values()], np.float64)}
    for n, v in grads.items():
        if v.numel() <= 4096:
            out["grad/" + n] = v
    save("training_grads_uvit.npz", xs=xs, k=k, masks=masks, poses=poses, digest=np.array(weights_digest(params)), **out)


def stochastic_samplers(R):
    """The stochastic sampling steps of the reference run by the reference itself (discrete_diffusion.py:423-452 ddpm_sample_step,
    :454-538 ddim_sample_step with eta > 0): full `_predict_videos` runs of the tiny pose model under vanilla History Guidance with
    every normal draw recorded.  (a) DDIM, eta = 0.5, 3 sampling steps of 1000; (b) DDPM: a 6-level schedule sampled with all 6
    levels (sampling_timesteps == timesteps selects ddpm_sample_step)."""
    A = R["AttrDict"]
    out = {}
    for tag, mod in (("eta", dict(ddim_sampling_eta=0.5, sampling_timesteps=3)), ("ddpm", dict(timesteps=6, sampling_timesteps=6))):
        cfg = algo_cfg(A, 16, TINY, sampling_steps=3, pred_hg=dict(name="vanilla", guidance_scale=4.0))
        for k, v in mod.items():
            cfg.diffusion[k] = v
        algo, _, p = build_algo(R, cfg)
        assert algo.diffusion_model.is_ddim_sampling == (tag == "eta")
        g = torch.Generator().manual_seed(61)
        vid = torch.randn(1, 8, 3, 16, 16, generator=g)
        cnd = synth_poses(1, 8, seed=9)
        algo.generator = torch.Generator().manual_seed(0)
        with RandnRecorder() as rec:
            res = algo._predict_videos(vid.clone(), n_context_tokens=1, conditions=cnd.clone())
        assert torch.isfinite(res).all()
        out.update({f"{tag}_xs": vid, f"{tag}_conds": cnd, f"{tag}_out": res, f"{tag}_n_noise": np.array(len(rec.draws))})
        out.update({f"{tag}_noise{i}": d for i, d in enumerate(rec.draws)})
        dm = algo.diffusion_model
        if tag == "ddpm":
            out.update(ddpm_alphas_cumprod=dm.alphas_cumprod, ddpm_coef1=dm.posterior_mean_coef1, ddpm_coef2=dm.posterior_mean_coef2,
                       ddpm_log_var=dm.posterior_log_variance_clipped)
        out["digest"] = np.array(weights_digest(p))
    save("sampler_stochastic.npz", **out)


def reconstruction_guidance_run(R):
    """sampler_recon.npz: the reference's own `_predict_videos` with cfg.diffusion.reconstruction_guidance > 0 (dfot_video.py:700-723 builds
    the guidance function, discrete_diffusion.py:485-513 differentiates the prediction w.r.t. x_t): tiny pose model, conditional history
    guidance (one branch), 2 context frames, 3 DDIM steps, every normal draw recorded; plus the same run without guidance, so that a
    test can see that the pull is far larger than its tolerance."""
    A = R["AttrDict"]
    out = {}
    for tag, w in (("rg", 400.0), ("plain", 0.0)):
        cfg = algo_cfg(A, 16, TINY, context=2, sampling_steps=3, pred_hg=dict(name="conditional"))
        cfg.diffusion["reconstruction_guidance"] = w
        algo, _, p = build_algo(R, cfg)
        g = torch.Generator().manual_seed(67)
        vid = torch.randn(1, 8, 3, 16, 16, generator=g)
        cnd = synth_poses(1, 8, seed=10)
        algo.generator = torch.Generator().manual_seed(0)
        with RandnRecorder() as rec:
            res = algo._predict_videos(vid.clone(), n_context_tokens=2, conditions=cnd.clone())
        assert torch.isfinite(res).all()
        out.update({f"{tag}_out": res.detach(), f"{tag}_n_noise": np.array(len(rec.draws))})
        out.update({f"{tag}_noise{i}": d for i, d in enumerate(rec.draws)})
        out.update(xs=vid, conds=cnd, weight=np.array(400.0), digest=np.array(weights_digest(p)))
    save("sampler_recon.npz", **out)


def vae_decode_golden():
    """The reference's own VideoVAE (default causal module choice, K600 latent geometry: 16 latent channels, 4x temporal / 8x spatial)
    with seeded random weights decodes seeded latents: `VideoVAE.decode(z, desired_length)` and the `_decode` convention `* 0.5 + 0.5`.
    Reduced size so that the CPU oracle test stays short: hidden 128 (widths 128/256/512/512 as the default), latent 3 x 16 x 8 -> 9
    frames of 128 x 64."""
    VideoVAE = ref_loader.install_vae()
    torch.manual_seed(0)
    vae = VideoVAE(hidden_size=128, z_channels=16, embed_dim=16, resolution=128, temporal_length=9, num_res_blocks=2).eval()
    from oracle import vae as ovae
    g = torch.Generator().manual_seed(71)
    sd = {n: ovae.seeded_tensor(n, t.shape) for n, t in vae.state_dict().items() if n.startswith(("decoder.", "post_quant_conv."))}
    missing, unexpected = vae.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith(("encoder.", "quant_conv.")) for m in missing)
    z = torch.randn(2, 16, 3, 16, 8, generator=g)
    with torch.no_grad():
        full = vae.decode(z)
        last = vae.decode(z, 7)
    assert full.shape == (2, 3, 9, 128, 64) and torch.equal(last, full[:, :, -7:])
    save("vae_decode.npz", z=z, frames=full, seed=np.array(71), names=np.array(sorted(sd)), shapes=np.array([str(tuple(sd[n].shape)) for n in sorted(sd)]))


@torch.no_grad()
def main():
    os.makedirs(OUT, exist_ok=True)
    if os.environ.get("ONLY") == "vae":
        return vae_decode_golden()
    R = ref_loader.install()
    A = R["AttrDict"]
    if os.environ.get("ONLY") == "training_grads_uvit":
        return training_grads_uvit(R)
    if os.environ.get("ONLY") == "stochastic":
        return stochastic_samplers(R)
    if os.environ.get("ONLY") == "recon":
        return reconstruction_guidance_run(R)

    # ---------------------------------------------------------------- schedule + scheduling matrices
    print("schedule")
    cfg = algo_cfg(A, 16, TINY)
    algo, ocfg, params = build_algo(R, cfg)
    dm = algo.diffusion_model
    sm_full = algo._generate_scheduling_matrix(8, 0)
    sm_pad = algo._generate_scheduling_matrix(5, 3)
    save("schedule.npz", alphas_cumprod=dm.alphas_cumprod, sqrt_alphas_cumprod=dm.sqrt_alphas_cumprod,
         sqrt_one_minus_alphas_cumprod=dm.sqrt_one_minus_alphas_cumprod, logsnr=dm.logsnr,
         ddim_levels=dm.ddim_idx_to_noise_level(torch.arange(51)), sched_8_0=sm_full, sched_5_3=sm_pad,
         train_t=torch.linspace(0, 1, 33), train_logsnr=dm.training_schedule(torch.linspace(0, 1, 33)))

    # other scheduling matrices of the reference (base_pytorch_video_algo.py:876-941), 7 sampling steps
    extra = {}
    cfg7 = algo_cfg(A, 16, TINY, sampling_steps=7)
    algo7, _, _ = build_algo(R, cfg7)
    for kind in ("interleaved", "gibbs", "autoregressive"):
        algo7.cfg["scheduling_matrix"] = kind
        extra[f"{kind}_8_0"] = algo7._generate_scheduling_matrix(8, 0)
        extra[f"{kind}_5_3"] = algo7._generate_scheduling_matrix(5, 3)
    cfg50 = algo_cfg(A, 16, TINY, sampling_steps=50)
    algo50, _, _ = build_algo(R, cfg50)
    extra["refine50_5_3"] = algo50._generate_refine_scheduling_matrix(horizon=5, goback_length=20, n_goback=2, padding=3)
    save("schedule_extra.npz", sampling_steps=np.array(7), **extra)

    # ---------------------------------------------------------------- ray encoding
    print("ray encoding")
    poses = synth_poses(2, 8, seed=1)
    cfg8 = algo_cfg(A, 8, TINY)
    algo8, _, _ = build_algo(R, cfg8)
    enc8 = algo8._process_conditions(poses.clone())
    cfg256 = algo_cfg(A, 256, TINY)
    algo256 = R["DFoTVideoPose"](cfg256)
    enc256 = algo256._process_conditions(poses[:1].clone())
    rows = [0, 1, 100, 255]
    save("ray_encoding.npz", poses=poses, enc8=enc8, rows256=np.array(rows), enc256_rows=enc256[:, :, :, rows, :])
    del algo256, enc256

    # ---------------------------------------------------------------- backbone, tiny
    print("backbone tiny")
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 8, 3, 16, 16, generator=g)
    kf = 0.125 * dm.logsnr[torch.randint(0, 1000, (2, 8), generator=g)]
    pz = synth_poses(2, 8, seed=2)
    cond = algo._process_conditions(pz.clone())
    mask = torch.tensor([True, False])
    v_masked = dm.model(x, kf, cond, mask)
    v_nomask = dm.model(x, kf, cond, None)
    save("backbone_tiny.npz", x=x, k=kf, poses=pz, mask=mask, v_masked=v_masked, v_nomask=v_nomask,
         digest=np.array(weights_digest(params)))

    # ---------------------------------------------------------------- backbone, RE10K widths, res 64
    print("backbone w64 (RE10K widths, resolution 64)")
    cfgw = algo_cfg(A, 64, W64)
    algow, ocfgw, paramsw = build_algo(R, cfgw, seed=3)
    g = torch.Generator().manual_seed(12)
    xw = torch.randn(2, 8, 3, 64, 64, generator=g)
    kw = 0.125 * dm.logsnr[torch.randint(0, 1000, (2, 8), generator=g)]
    pw = synth_poses(2, 8, seed=4)
    condw = algow._process_conditions(pw.clone())
    vw = algow.diffusion_model.model(xw, kw, condw, torch.tensor([True, False]))
    save("backbone_w64.npz", x=xw, k=kw, poses=pw, mask=np.array([True, False]), v=vw,
         digest=np.array(weights_digest(paramsw)))
    del algow, paramsw

    # ---------------------------------------------------------------- single step traces per scheme
    print("step traces")
    schemes = {
        "conditional": dict(name="conditional"),
        "vanilla": dict(name="vanilla", guidance_scale=4.0),
        "stabilized_vanilla": dict(name="stabilized_vanilla", guidance_scale=4.0, stabilization_level=0.02),
        "fractional": dict(name="fractional", guidance_scale=3.0, freq_scale=0.4),
    }
    HG = R["HistoryGuidance"]
    step = {}
    g = torch.Generator().manual_seed(21)
    xs = torch.randn(2, 8, 3, 16, 16, generator=g)
    conds = synth_poses(2, 8, seed=5)
    cmask = torch.tensor([[1, 2, 2, 0, 0, 0, 0, -1]] * 2)
    frm = torch.tensor([[-1, -1, -1, 499, 499, 499, 499, 999]] * 2)
    to = torch.tensor([[-1, -1, -1, 479, 479, 479, 479, 999]] * 2)
    step.update(xs=xs, conds=conds, cmask=cmask, frm=frm, to=to)
    for sname, sc in schemes.items():
        hgo = HG.from_config(A(dict(sc, visualize=False)), timesteps=1000)
        with RandnRecorder() as rec:
            with hgo(cmask) as mgr:
                xi, fi, ti, cm = mgr.prepare(xs.clone(), frm.clone(), to.clone(), replacement_fn=dm.q_sample,
                                             replacement_only=False)
                nfe = mgr.nfe
                cc = algo._process_conditions(conds.repeat_interleave(nfe, 0).clone(), fi)
                xo = dm.sample_step(xi, fi, ti, cc, cm)
                xc = mgr.compose(xo)
        step[f"{sname}_nfe"] = np.array(nfe)
        step[f"{sname}_x_in"] = xi
        step[f"{sname}_from"] = fi
        step[f"{sname}_to"] = ti
        step[f"{sname}_cond_mask"] = np.array([]) if cm is None else cm
        step[f"{sname}_x_out"] = xo
        step[f"{sname}_x_composed"] = xc
        for i, d in enumerate(rec.draws):
            step[f"{sname}_noise{i}"] = d
        step[f"{sname}_n_noise"] = np.array(len(rec.draws))
    save("step_trace.npz", digest=np.array(weights_digest(params)), **step)

    # ---------------------------------------------------------------- sampler, 8 frames
    print("sampler 8f")
    cfg8f = algo_cfg(A, 16, TINY, sampling_steps=3, pred_hg=dict(name="vanilla", guidance_scale=4.0))
    algo8f, _, p8 = build_algo(R, cfg8f)
    g = torch.Generator().manual_seed(31)
    vid = torch.randn(1, 8, 3, 16, 16, generator=g)
    cnd = synth_poses(1, 8, seed=6)
    algo8f.generator = torch.Generator().manual_seed(0)
    with RandnRecorder() as rec:
        out = algo8f._predict_videos(vid.clone(), n_context_tokens=1, conditions=cnd.clone())
    arrays = {f"noise{i}": d for i, d in enumerate(rec.draws)}
    save("sampler_8f.npz", xs=vid, conds=cnd, out=out, n_noise=np.array(len(rec.draws)),
         digest=np.array(weights_digest(p8)), **arrays)

    # ---------------------------------------------------------------- sampler, 200 frames
    print("sampler 200f")
    cfg200 = algo_cfg(A, 16, TINY, n_frames=200, sampling_steps=2,
                      pred_hg=dict(name="stabilized_vanilla", guidance_scale=4.0, stabilization_level=0.02),
                      interp_hg=dict(name="vanilla", guidance_scale=1.5), density=0.0625, interp_enabled=True,
                      max_batch=4)
    algo200, _, p200 = build_algo(R, cfg200)
    g = torch.Generator().manual_seed(41)
    vid = torch.randn(1, 200, 3, 16, 16, generator=g)
    cnd = synth_poses(1, 200, seed=7)
    algo200.generator = torch.Generator().manual_seed(0)
    torch.manual_seed(777)  # global stream feeds every un-seeded randn_like of the reference
    calls = []
    orig = algo200._sample_sequence

    def spy(batch_size, length=None, context=None, context_mask=None, **kw):
        calls.append((batch_size, context_mask.clone()))
        return orig(batch_size, length=length, context=context, context_mask=context_mask, **kw)
    algo200._sample_sequence = spy
    with RandnRecorder() as rec:
        out = algo200._predict_videos(vid.clone(), n_context_tokens=1, conditions=cnd.clone())
    # noise is NOT stored: the test re-draws it (init: Generator(0); everything else: global seed 777,
    # same call order and shapes), only the shapes of the draws are kept as a cross-check
    draw_shapes = np.array([list(d.shape) + [0] * (6 - d.ndim) for d in rec.draws])
    call_batches = np.array([c[0] for c in calls])
    call_masks = np.concatenate([np.pad(c[1].numpy(), ((0, 0), (0, 8 - c[1].shape[1])), constant_values=-1)
                                 for c in calls], 0)
    save("sampler_200f.npz", xs=vid, conds=cnd, out=out, n_noise=np.array(len(rec.draws)),
         draw_shapes=draw_shapes, call_batches=call_batches, call_masks=call_masks,
         digest=np.array(weights_digest(p200)))

    # ---------------------------------------------------------------- training loss
    print("training loss")
    g = torch.Generator().manual_seed(51)
    xt = torch.randn(2, 8, 3, 16, 16, generator=g)
    tt = torch.rand(2, 8, generator=g)
    pc = synth_poses(2, 8, seed=8)
    cc = algo._process_conditions(pc.clone())
    with RandnRecorder() as rec:
        x_pred, loss = dm(xt, cc, tt)
    save("training_loss.npz", x=xt, t=tt, poses=pc, noise=rec.draws[0], x_pred=x_pred, loss=loss,
         digest=np.array(weights_digest(params)))
    print("done")


if __name__ == "__main__":
    main()
