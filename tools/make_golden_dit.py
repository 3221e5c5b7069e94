#!/usr/bin/env python3
"""Generate the DiT3D / Kinetics-600 fixtures under tests/golden/ by executing the reference's own source on CPU.

Run ONLY in the build container (needs /root/reference):   python tools/make_golden_dit.py
Fixtures are data (inputs, injected noise, expected outputs); weights come from oracle.dit.seeded_params(cfg, seed),
loaded strictly into the reference module and re-created bit-identically by the tests (sha256 stored).

  dit_tiny.npz      DiT3D.forward: hidden 128 / depth 3 / 4 heads (head dim 32 -> uneven RoPE split 12/10/10), 16x8 latents,
                    T=5 and a shorter T=3 window; a second model (hidden 192, 6 heads) with patch 2 on 32x16 and
                    spatial_mlp_ratio 4 (MLP branch)
  dit_k600.npz      DiT3D.forward at the K600 size: DiT/XL (hidden 1152, depth 28, 16 heads), latents 16x16x16, T=5
  diffdit.npz       DifferenceDiT3D.forward (factorized_matrix_attention, sinusoidal_2d, interleaved): a tiny model (hidden 128,
                    depth 2, E 64, 1x4 matrix heads), a tiny model with 2 column heads and no biases / no temporal MLP, and the
                    bash/k600 width (hidden 1152, 12 spatial heads, E 64, 1x16 matrix heads, MLP ratios 4) at depth 3
  sampler_k600_diff.npz  DifferenceDFoTVideo: torch.diff + merge_tensors -> _predict_videos on the 10 merged tokens (context 2 frames =
                    4 merged tokens, 3 DDIM steps, vanilla history guidance 1.5, tiny difference model) -> unmerge_tensors
  discrete_loss.npz DiscreteDiffusion.forward (pred_v, cosine) on the small DiT with injected noise: x_pred, weighted loss, and
                    compute_loss_weights for fused_min_snr (decay 0.96 / 0.9), min_snr, sigmoid, uniform
  hg_temporal.npz   temporal / custom History Guidance (history sub-sequences, several gen segments) on the small DiT: one
                    prepare -> sample_step -> compose per scheme, and DFoTVideo._predict_videos with the temporal scheme
  training_noise.npz  BaseVideoAlgo._get_training_noise_levels (continuous RE10K-style algo and discrete K600-style algo) for
                    random_independent / random_uniform / interleaved / uniform_future / fixed_context / variable_context,
                    generator seed 123
  sampler_refine.npz  DFoTVideo._sample_sequence_refine (refinement ladder: DDIM steps + q_sample_from_x_k re-noising), small DiT,
                    RE10K schedule, 6 DDIM indices, goback_length 2, n_goback 2, conditional guidance (+ the fact that the
                    reference returns NaN for a padded window)
  training_grads.npz  gradients of the reference's own training loss: DFoTVideo (small DiT, cosine / pred_v / fused_min_snr) and
                    DifferenceDFoTVideo (tiny difference model): diffusion_model(xs, None, k) with recorded noise -> _reweight_loss with
                    masks -> backward(); stored: the loss, the L2 norm of every parameter gradient and the full gradient of a few tensors
  sampler_k600.npz  DFoTVideo._predict_videos with DiscreteDiffusion (cosine, pred_v, integer levels): 5 tokens,
                    context 2, 4 DDIM steps, vanilla history guidance 2.0 (small DiT) with the injected noise
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ref_loader  # noqa: E402
from make_golden import RandnRecorder, save, weights_digest  # noqa: E402
from oracle import dit as odit  # noqa: E402

torch.set_num_threads(8)


def ref_dit(R, ocfg: odit.DiTConfig, seed: int):
    A = R["AttrDict"]
    cfg = A(dict(name="dit3d", variant="full", pos_emb_type="rope_3d", patch_size=ocfg.patch_size,
                 hidden_size=ocfg.hidden_size, depth=ocfg.depth, num_heads=ocfg.num_heads, mlp_ratio=4.0,
                 use_gradient_checkpointing=False,
                 **({"spatial_mlp_ratio": ocfg.spatial_mlp_ratio} if ocfg.spatial_mlp_ratio else {})))
    m = R["DiT3D"](cfg, x_shape=[ocfg.in_channels, *ocfg.resolution], max_tokens=ocfg.max_tokens,
                   external_cond_type="action", external_cond_num_classes=None, external_cond_dim=0,
                   use_causal_mask=False).eval()
    params = odit.seeded_params(ocfg, seed)
    assert list(m.state_dict().keys()) == list(params.keys()), "oracle parameter inventory differs from the reference"
    m.load_state_dict(params, strict=True)
    return m, params


def ref_diffdit(R, ocfg: "odit.DiffDiTConfig", seed: int):
    import importlib
    A = R["AttrDict"]
    dd = importlib.import_module("algorithms.dfot.backbones.dit.difference_dit3d")
    cfg = A(dict(name="difference_dit3d", variant="factorized_matrix_attention", pos_emb_type="sinusoidal_2d", merge_type="interleaved",
                 patch_size=ocfg.patch_size, hidden_size=None, embed_col_dim=ocfg.embed_col_dim, embed_row_dim=ocfg.hidden_size,
                 num_heads=ocfg.num_heads, num_col_heads=ocfg.num_col_heads, num_row_heads=ocfg.num_row_heads, depth=ocfg.depth,
                 mlp_ratio=ocfg.mlp_ratio or None, spatial_mlp_ratio=ocfg.spatial_mlp_ratio, use_bias=ocfg.use_bias, matrix_block="matrix",
                 flatten_matrix_rope=False, matrix_multi_token=False, use_gradient_checkpointing=False))
    m = dd.DifferenceDiT3D(cfg, x_shape=[ocfg.in_channels, *ocfg.resolution], max_tokens=ocfg.max_tokens, external_cond_type="action",
                           external_cond_num_classes=None, external_cond_dim=0, use_causal_mask=False).eval()
    params = odit.diff_seeded_params(ocfg, seed)
    assert list(m.state_dict().keys()) == list(params.keys()), "oracle parameter inventory differs from the reference"
    m.load_state_dict(params, strict=True)
    return m, params


DIFF_TINY = dict(hidden_size=128, depth=2, num_heads=4, in_channels=4, resolution=(16, 8), embed_col_dim=64, num_row_heads=4)
DIFF_TINY2 = dict(hidden_size=128, depth=1, num_heads=2, in_channels=4, resolution=(16, 8), embed_col_dim=64, num_col_heads=2,
                  num_row_heads=2, use_bias=False, mlp_ratio=0.0)
DIFF_WIDE = dict(depth=3)


def video_cfg(A, ocfg: odit.DiTConfig, sampling_steps: int, hg: dict):
    c = ocfg.in_channels
    return A(dict(
        debug=False, lr=5e-5, x_shape=[c, *ocfg.resolution], max_frames=ocfg.max_tokens, n_frames=ocfg.max_tokens,
        frame_skip=1, context_frames=2,
        latent=dict(enabled=False, type="pre_sample", suffix=None, downsampling_factor=[1, 1], shape=None, num_channels=c),
        data_mean=[[[0.0]]] * c, data_std=[[[1.0]]] * c,
        external_cond_type="action", external_cond_num_classes=None, external_cond_dim=0, external_cond_stack=False,
        external_cond_processing=None, compile=False, weight_decay=0.0, optimizer_beta=[0.9, 0.99],
        lr_scheduler=dict(name="constant_with_warmup", num_warmup_steps=10),
        noise_level="random_independent", uniform_future=dict(enabled=False),
        fixed_context=dict(enabled=False, indices=None, dropout=0),
        variable_context=dict(enabled=False, prob=0.25, dropout=0.3),
        chunk_size=-1, scheduling_matrix="full_sequence", replacement="noisy_scale",
        refinement_sampling=dict(enabled=False, goback_length=20, n_goback=5),
        save_attn_map=dict(enabled=False, attn_map_dir=None),
        diffusion=dict(is_continuous=False, timesteps=1000, beta_schedule="cosine", schedule_fn_kwargs=dict(shift=1.0),
                       use_causal_mask=False, clip_noise=20.0, objective="pred_v",
                       loss_weighting=dict(strategy="fused_min_snr", snr_clip=5.0, cum_snr_decay=0.96),
                       sampling_timesteps=sampling_steps, ddim_sampling_eta=0.0, reconstruction_guidance=0.0),
        vae=dict(pretrained_path=None, pretrained_kwargs={}, use_fp16=False, batch_size=2),
        checkpoint=dict(reset_optimizer=False, strict=True),
        tasks=dict(prediction=dict(enabled=True, history_guidance=dict(hg, visualize=False), keyframe_density=None,
                                   sliding_context_len=None),
                   interpolation=dict(enabled=False, history_guidance=dict(name="conditional", visualize=False),
                                      max_batch_size=None)),
        logging=dict(deterministic=0, loss_freq=100, grad_norm_freq=100, max_num_videos=8, n_metrics_frames=None,
                     metrics=[], metrics_batch_size=16, sanity_generation=False, raw_dir=None),
        backbone=dict(name="dit3d", variant="full", pos_emb_type="rope_3d", patch_size=ocfg.patch_size,
                      hidden_size=ocfg.hidden_size, depth=ocfg.depth, num_heads=ocfg.num_heads, mlp_ratio=4.0,
                      use_gradient_checkpointing=False),
    ))


HG_TEMPORAL = {
    "temporal": dict(name="temporal", hist_subsequences=[[0], [1], [0, 1]], hist_weights=[0.5, 0.5, 1.0], gen_segments=[[0, 1], [1, 2]]),
    "custom": dict(name="custom", hist_segments=[dict(time_indices=[0, -1], freq_ranges=[[0.0, 1.0], [0.3, 1.0]],
                                                      freq_ranges_if_generated=[[0.1, 1.0]])],
                   hist_weights=[2.0], gen_segments=None),
}


TRAIN_NOISE_CASES = {
    "indep": dict(noise_level="random_independent"),
    "uniform": dict(noise_level="random_uniform"),
    "interleaved": dict(noise_level="interleaved"),
    "ufuture": dict(noise_level="random_independent", uniform_future=True),
    "fixed": dict(noise_level="random_independent", fixed=dict(enabled=True, indices=None, dropout=0.5)),
    "variable": dict(noise_level="random_uniform", variable=dict(enabled=True, prob=0.25, dropout=0.3)),
}


@torch.no_grad()
def training_noise_fixture(R):
    import types
    import make_golden as MG
    print("training noise levels")
    A = R["AttrDict"]
    algo_c, _, _ = MG.build_algo(R, MG.algo_cfg(A, 16, MG.TINY))  # continuous, 8 tokens, 1 context token
    small = odit.DiTConfig(hidden_size=128, depth=2, num_heads=4, patch_size=1, in_channels=4, resolution=(16, 8), max_tokens=5)
    algo_d = R["DFoTVideo"](video_cfg(A, small, sampling_steps=4, hg=dict(name="conditional"))).eval()  # discrete, 5 tokens, 2 context
    masks8 = torch.ones(3, 8, dtype=torch.bool)
    masks8[1, 6:] = False
    masks5 = torch.ones(3, 5, dtype=torch.bool)
    masks5[2, 4:] = False
    out = dict(masks8=masks8, masks5=masks5)
    for tag, algo, masks, nt in (("c", algo_c, masks8, 8), ("d", algo_d, masks5, 5)):
        algo.trainer = types.SimpleNamespace(training=True)
        for name, c in TRAIN_NOISE_CASES.items():
            algo.cfg["noise_level"] = c["noise_level"]
            algo.cfg["uniform_future"] = A(dict(enabled=bool(c.get("uniform_future"))))
            algo.cfg["fixed_context"] = A(c.get("fixed", dict(enabled=False, indices=None, dropout=0)))
            algo.cfg["variable_context"] = A(c.get("variable", dict(enabled=False, prob=0.25, dropout=0.3)))
            algo.generator = torch.Generator().manual_seed(123)
            lv, mk = algo._get_training_noise_levels(torch.zeros(3, nt, 4, 2, 2), masks.clone())
            out[f"{tag}_{name}_levels"], out[f"{tag}_{name}_masks"] = lv, mk
        out[f"{tag}_n_context"] = np.array(algo.n_context_tokens)
    save("training_noise.npz", **out)


@torch.no_grad()
def hg_temporal_fixture(R):
    print("history guidance: temporal / custom")
    A = R["AttrDict"]
    small = odit.DiTConfig(hidden_size=128, depth=2, num_heads=4, patch_size=1, in_channels=4, resolution=(16, 8), max_tokens=5)
    cfg = video_cfg(A, small, sampling_steps=3, hg=HG_TEMPORAL["temporal"])
    algo = R["DFoTVideo"](cfg).eval()
    ps = odit.seeded_params(small, 4)
    algo.diffusion_model.model.load_state_dict(ps, strict=True)
    dm = algo.diffusion_model
    g = torch.Generator().manual_seed(12)
    xs = torch.randn(2, 5, 4, 16, 8, generator=g)
    cmask = torch.tensor([[1, 2, 0, 0, 0]] * 2)
    frm = torch.tensor([[-1, -1, 499, 499, 499]] * 2)
    to = torch.tensor([[-1, -1, 479, 479, 479]] * 2)
    out = dict(xs=xs, cmask=cmask, frm=frm, to=to, digest=np.array(weights_digest(ps)))
    for sname, sc in HG_TEMPORAL.items():
        hgo = R["HistoryGuidance"].from_config(A(dict({k: v for k, v in sc.items() if v is not None}, visualize=False)), timesteps=1000)
        with RandnRecorder() as rec:
            with hgo(cmask) as mgr:
                xi, fi, ti, cm = mgr.prepare(xs.clone(), frm.clone(), to.clone(), replacement_fn=dm.q_sample, replacement_only=False)
                xo = dm.sample_step(xi, fi, ti, None, cm)
                xc = mgr.compose(xo)
            out[f"{sname}_nfe"] = np.array(mgr.nfe)
        out.update({f"{sname}_x_in": xi, f"{sname}_from": fi, f"{sname}_to": ti, f"{sname}_x_out": xo, f"{sname}_x_composed": xc,
                    f"{sname}_n_noise": np.array(len(rec.draws))})
        for i, d in enumerate(rec.draws):
            out[f"{sname}_noise{i}"] = d
    vid = torch.randn(2, 5, 4, 16, 8, generator=g)
    algo.generator = torch.Generator().manual_seed(0)
    with RandnRecorder() as rec:
        pred = algo._predict_videos(vid.clone(), n_context_tokens=2, conditions=None)
    out.update(vid=vid, pred=pred, pred_n_noise=np.array(len(rec.draws)))
    for i, d in enumerate(rec.draws):
        out[f"pred_noise{i}"] = d
    save("hg_temporal.npz", **out)


@torch.enable_grad()
def training_grads_fixture(R):
    print("training grads")
    A = R["AttrDict"]
    out = {}
    small = odit.DiTConfig(hidden_size=128, depth=2, num_heads=4, patch_size=1, in_channels=4, resolution=(16, 8), max_tokens=5)
    algo = R["DFoTVideo"](video_cfg(A, small, sampling_steps=4, hg=dict(name="conditional"))).train()
    ps = odit.seeded_params(small, 2)
    algo.diffusion_model.model.load_state_dict(ps, strict=True)
    oc = odit.DiffDiTConfig(**DIFF_TINY)
    cfg = video_cfg(A, small, sampling_steps=3, hg=dict(name="conditional"))
    cfg["backbone"] = A(dict(name="difference_dit3d", variant="factorized_matrix_attention", pos_emb_type="sinusoidal_2d",
                             merge_type="interleaved", patch_size=1, hidden_size=None, embed_col_dim=oc.embed_col_dim,
                             embed_row_dim=oc.hidden_size, num_heads=oc.num_heads, num_col_heads=1, num_row_heads=oc.num_row_heads,
                             depth=oc.depth, mlp_ratio=4.0, spatial_mlp_ratio=4.0, use_bias=True, matrix_block="matrix",
                             flatten_matrix_rope=False, matrix_multi_token=False, use_gradient_checkpointing=False))
    dalgo = R["DifferenceDFoTVideo"](cfg).train()
    dps = odit.diff_seeded_params(oc, 3)
    dalgo.diffusion_model.model.load_state_dict(dps, strict=True)
    g = torch.Generator().manual_seed(21)
    xs = torch.randn(2, 5, 4, 16, 8, generator=g)
    k = torch.randint(0, 1000, (2, 5), generator=g)
    masks = torch.ones(2, 5)
    masks[1, 3] = 0
    for tag, al, weights in (("dit", algo, ps), ("diff", dalgo, dps)):
        model = al.diffusion_model.model
        for p_ in model.parameters():
            p_.grad = None
        if tag == "diff":  # DifferenceDFoTVideo.training_step: differences interleaved with the frames, doubled levels and masks
            x_in = al.merge_tensors(torch.diff(xs, dim=1, prepend=xs[:, :1]), xs)
            k_in, m_in = al.merge_tensors(k, k), al.merge_tensors(masks, masks)
        else:
            x_in, k_in, m_in = xs, k, masks
        with RandnRecorder() as rec:
            _, loss = al.diffusion_model(x_in, None, k=k_in)
        loss = al._reweight_loss(loss, m_in)
        loss.backward()
        grads = {n: p_.grad.detach().clone() for n, p_ in model.named_parameters()}
        out[f"{tag}_loss"] = loss.detach()
        out[f"{tag}_noise"] = rec.draws[0]
        out[f"{tag}_names"] = np.array(list(grads))
        out[f"{tag}_norms"] = np.array([float(v.norm()) for v in grads.values()], np.float64)
        for n, v in grads.items():
            if v.numel() <= 4096 or n.endswith(("attn.qkv_u", "attn.proj_u")):
                out[f"{tag}_grad/{n}"] = v
        out[f"{tag}_digest"] = np.array(weights_digest(weights))
    save("training_grads.npz", xs=xs, k=k, masks=masks, **out)


@torch.no_grad()
def refine_fixture(R):
    print("sampler refine")
    small = odit.DiTConfig(hidden_size=128, depth=2, num_heads=4, patch_size=1, in_channels=4, resolution=(16, 8), max_tokens=5)
    # the K600 cosine schedule has alphas_cumprod[T-1] == 0 in fp32: q_sample_from_x_k then divides 0/0 for every context token
    # (their level -1 gathers the LAST table entry) and the reference returns NaN everywhere -- the fixture therefore uses the
    # RE10K schedule (cosine_simple_diffusion shifted 0.125, alphas_cumprod[T-1] > 0) with the discrete model
    vc = video_cfg(R["AttrDict"], small, sampling_steps=6, hg=dict(name="conditional"))
    vc["diffusion"]["beta_schedule"] = "cosine_simple_diffusion"
    vc["diffusion"]["schedule_fn_kwargs"] = R["AttrDict"](dict(shifted=0.125, interpolated=False))
    algo = R["DFoTVideo"](vc).eval()
    assert float(algo.diffusion_model.alphas_cumprod[-1]) > 0
    ps = odit.seeded_params(small, 2)
    algo.diffusion_model.model.load_state_dict(ps, strict=True)
    g = torch.Generator().manual_seed(12)
    vid = torch.randn(2, 5, 4, 16, 8, generator=g)
    mask = torch.tensor([[1, 1, 0, 0, 0]] * 2)
    algo.generator = torch.Generator().manual_seed(0)
    with RandnRecorder() as rec:
        out, _ = algo._sample_sequence_refine(2, goback_length=2, n_goback=2, context=vid.clone(), context_mask=mask.clone())
    arrays = {f"noise{i}": d for i, d in enumerate(rec.draws)}
    # a shorter window: the padded last column stays at pure noise, so the reference treats EVERY row as a re-noising row, also
    # the descending ones, where alphas_cumprod[to] / alphas_cumprod[from] > 1 and sqrt(1 - scale) is NaN: recorded as a fact only
    out4, _ = algo._sample_sequence_refine(2, goback_length=2, n_goback=2, context=vid[:, :4].clone(), context_mask=mask[:, :4].clone())
    assert torch.isfinite(out).all()
    save("sampler_refine.npz", xs=vid, mask=mask, out=out, padded_window_is_nan=np.array(bool(torch.isnan(out4).any())),
         n_noise=np.array(len(rec.draws)), digest=np.array(weights_digest(ps)), **arrays)


@torch.no_grad()
def diff_sampler_fixture(R):
    print("sampler k600 difference")
    A = R["AttrDict"]
    oc = odit.DiffDiTConfig(**DIFF_TINY)
    base = odit.DiTConfig(hidden_size=128, depth=2, num_heads=4, patch_size=1, in_channels=4, resolution=(16, 8), max_tokens=5)
    cfg = video_cfg(A, base, sampling_steps=3, hg=dict(name="vanilla", guidance_scale=1.5))
    cfg["backbone"] = A(dict(name="difference_dit3d", variant="factorized_matrix_attention", pos_emb_type="sinusoidal_2d",
                             merge_type="interleaved", patch_size=1, hidden_size=None, embed_col_dim=oc.embed_col_dim,
                             embed_row_dim=oc.hidden_size, num_heads=oc.num_heads, num_col_heads=1, num_row_heads=oc.num_row_heads,
                             depth=oc.depth, mlp_ratio=4.0, spatial_mlp_ratio=4.0, use_bias=True, matrix_block="matrix",
                             flatten_matrix_rope=False, matrix_multi_token=False, use_gradient_checkpointing=False))
    algo = R["DifferenceDFoTVideo"](cfg).eval()
    ps = odit.diff_seeded_params(oc, 3)
    algo.diffusion_model.model.load_state_dict(ps, strict=True)
    g = torch.Generator().manual_seed(9)
    vid = torch.randn(2, 5, 4, 16, 8, generator=g)
    merged = algo.merge_tensors(torch.diff(vid, dim=1, prepend=vid[:, :1]), vid)
    algo.generator = torch.Generator().manual_seed(0)
    with RandnRecorder() as rec:
        out = algo._predict_videos(merged.clone(), n_context_tokens=4, conditions=None)
    gen_diff, gen = algo.unmerge_tensors(out)
    arrays = {f"noise{i}": d for i, d in enumerate(rec.draws)}
    save("sampler_k600_diff.npz", xs=vid, merged=merged, out=out, gen=gen, gen_diff=gen_diff, n_noise=np.array(len(rec.draws)),
         digest=np.array(weights_digest(ps)), **arrays)


def recon_k600_fixture(R):
    """sampler_recon_k600.npz: the reference's own `DFoTVideo._predict_videos` on the DISCRETE cosine schedule with
    cfg.diffusion.reconstruction_guidance > 0 (dfot_video.py:700-723, discrete_diffusion.py:485-513).  The first DDIM step sits at
    the zero-terminal-SNR level (alphas_cumprod = 0), where the reference keeps the unguided x0 but still shifts the predicted noise:
    the branch the continuous fixture (sampler_recon.npz) never reaches.  Tiny DiT3D, conditional (one-branch) guidance, 2 context
    tokens, 3 DDIM steps, every normal draw recorded; plus the same run without guidance."""
    print("reconstruction guidance, discrete cosine schedule")
    A = R["AttrDict"]
    small = odit.DiTConfig(hidden_size=128, depth=2, num_heads=4, patch_size=1, in_channels=4, resolution=(16, 8), max_tokens=5)
    ps = odit.seeded_params(small, 2)
    g = torch.Generator().manual_seed(19)
    vid = torch.randn(2, 5, 4, 16, 8, generator=g)
    out = dict(xs=vid, weight=np.array(3.0e4), digest=np.array(weights_digest(ps)))
    for tag, w in (("rg", 3.0e4), ("plain", 0.0)):
        cfg = video_cfg(A, small, sampling_steps=3, hg=dict(name="conditional"))
        cfg.diffusion["reconstruction_guidance"] = w
        algo = R["DFoTVideo"](cfg).eval()
        algo.diffusion_model.model.load_state_dict(ps, strict=True)
        algo.generator = torch.Generator().manual_seed(0)
        with RandnRecorder() as rec:
            res = algo._predict_videos(vid.clone(), n_context_tokens=2, conditions=None)
        assert torch.isfinite(res).all()
        out.update({f"{tag}_out": res.detach(), f"{tag}_n_noise": np.array(len(rec.draws))})
        out.update({f"{tag}_noise{i}": d for i, d in enumerate(rec.draws)})
    save("sampler_recon_k600.npz", **out)


def main():
    R = ref_loader.install()
    A = R["AttrDict"]
    if os.environ.get("ONLY") == "recon_k600":
        return recon_k600_fixture(R)  # (differentiates the prediction w.r.t. x_t: not under no_grad)
    with torch.no_grad():
        return _main_no_grad(R, A)


def _main_no_grad(R, A):
    if os.environ.get("ONLY") == "refine":
        return refine_fixture(R)
    if os.environ.get("ONLY") == "training_grads":
        return training_grads_fixture(R)

    print("dit tiny")
    tiny = odit.DiTConfig(hidden_size=128, depth=3, num_heads=4, patch_size=1, in_channels=4, resolution=(16, 8), max_tokens=5)
    m, p = ref_dit(R, tiny, 0)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 5, 4, 16, 8, generator=g)
    k = torch.randint(0, 1000, (2, 5), generator=g)
    tiny_mlp = odit.DiTConfig(hidden_size=192, depth=2, num_heads=6, patch_size=2, in_channels=4, resolution=(32, 16),
                              max_tokens=5, spatial_mlp_ratio=4.0)
    m2, p2 = ref_dit(R, tiny_mlp, 1)
    x2 = torch.randn(2, 5, 4, 32, 16, generator=g)
    save("dit_tiny.npz", x=x, k=k, out=m(x, k), out_t3=m(x[:, :3], k[:, :3]), digest=np.array(weights_digest(p)),
         x_mlp=x2, out_mlp=m2(x2, k), digest_mlp=np.array(weights_digest(p2)))

    print("dit k600")
    xl = odit.DiTConfig()
    m, p = ref_dit(R, xl, 0)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 5, 16, 16, 16, generator=g)
    k = torch.tensor([[0, 17, 500, 871, 999]])
    hooks, blocks = [], {}
    for i in (0, 13, 27):
        hooks.append(m.dit_base.blocks[i].register_forward_hook(lambda mod, a, o, i=i: blocks.__setitem__(i, o.clone())))
    out = m(x, k)
    # per-block fixtures: mean |x| and the first 64 channels of 4 tokens (the residual stream is [1, 1280, 1152])
    extra = {}
    for i, o in blocks.items():
        extra[f"block{i}_absmean"] = o.abs().mean()
        extra[f"block{i}_rows"] = o[0, [0, 255, 700, 1279], :64]
    save("dit_k600.npz", x=x, k=k, out=out, digest=np.array(weights_digest(p)), **extra)
    del m

    print("difference dit")
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 10, 4, 16, 8, generator=g)
    k = torch.randint(0, 1000, (2, 10), generator=g)
    c1, c2, c3 = odit.DiffDiTConfig(**DIFF_TINY), odit.DiffDiTConfig(**DIFF_TINY2), odit.DiffDiTConfig(**DIFF_WIDE)
    m1, p1 = ref_diffdit(R, c1, 0)
    m2, p2 = ref_diffdit(R, c2, 1)
    m3, p3 = ref_diffdit(R, c3, 2)
    xw = torch.randn(1, 10, 16, 16, 16, generator=g)
    kw = torch.tensor([[0, 0, 17, 17, 500, 500, 871, 871, 999, 999]])
    save("diffdit.npz", x=x, k=k, out=m1(x, k), out_t6=m1(x[:, :6], k[:, :6]), digest=np.array(weights_digest(p1)),
         out2=m2(x, k), digest2=np.array(weights_digest(p2)), xw=xw, kw=kw, outw=m3(xw, kw), digestw=np.array(weights_digest(p3)))
    del m1, m2, m3

    print("sampler k600")
    small = odit.DiTConfig(hidden_size=128, depth=2, num_heads=4, patch_size=1, in_channels=4, resolution=(16, 8), max_tokens=5)
    cfg = video_cfg(A, small, sampling_steps=4, hg=dict(name="vanilla", guidance_scale=2.0))
    algo = R["DFoTVideo"](cfg).eval()
    ps = odit.seeded_params(small, 2)
    algo.diffusion_model.model.load_state_dict(ps, strict=True)
    dm = algo.diffusion_model
    g = torch.Generator().manual_seed(7)
    vid = torch.randn(2, 5, 4, 16, 8, generator=g)
    algo.generator = torch.Generator().manual_seed(0)
    with RandnRecorder() as rec:
        out = algo._predict_videos(vid.clone(), n_context_tokens=2, conditions=None)
    arrays = {f"noise{i}": d for i, d in enumerate(rec.draws)}
    save("sampler_k600.npz", xs=vid, out=out, n_noise=np.array(len(rec.draws)), digest=np.array(weights_digest(ps)),
         alphas_cumprod=dm.alphas_cumprod, sqrt_alphas_cumprod=dm.sqrt_alphas_cumprod,
         sqrt_one_minus_alphas_cumprod=dm.sqrt_one_minus_alphas_cumprod, **arrays)
    refine_fixture(R)
    training_grads_fixture(R)
    diff_sampler_fixture(R)
    hg_temporal_fixture(R)
    training_noise_fixture(R)
    print("discrete loss")
    g = torch.Generator().manual_seed(10)
    xt = torch.randn(2, 5, 4, 16, 8, generator=g)
    kt = torch.randint(0, 1000, (2, 5), generator=g)
    kt[0, 0], kt[1, 4] = 0, 999
    with RandnRecorder() as rec:
        x_pred, loss = dm(xt, None, kt)
    weights = {"w_fused_096": dm.compute_loss_weights(kt, "fused_min_snr")}
    dm.loss_weighting["cum_snr_decay"] = 0.9
    weights["w_fused_090"] = dm.compute_loss_weights(kt, "fused_min_snr")
    weights["w_min_snr"] = dm.compute_loss_weights(kt, "min_snr")
    weights["w_uniform"] = dm.compute_loss_weights(kt, "uniform").float()
    save("discrete_loss.npz", x=xt, k=kt, noise=rec.draws[0], x_pred=x_pred, loss=loss, digest=np.array(weights_digest(ps)), **weights)
    print("done")


if __name__ == "__main__":
    main()
