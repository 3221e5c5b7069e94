"""CPU: the DiT3D / Kinetics-600 oracle (oracle/dit.py + discrete-level sampler) against fixtures produced by running the
reference's own DiT3D / DFoTVideo / DiscreteDiffusion source (tools/make_golden_dit.py)."""
import hashlib
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import dit as odit
from oracle import sampler as osm
from oracle import schedule as sch

torch.set_num_threads(8)

TINY = odit.DiTConfig(hidden_size=128, depth=3, num_heads=4, patch_size=1, in_channels=4, resolution=(16, 8), max_tokens=5)
TINY_MLP = odit.DiTConfig(hidden_size=192, depth=2, num_heads=6, patch_size=2, in_channels=4, resolution=(32, 16),
                          max_tokens=5, spatial_mlp_ratio=4.0)
SMALL = odit.DiTConfig(hidden_size=128, depth=2, num_heads=4, patch_size=1, in_channels=4, resolution=(16, 8), max_tokens=5)


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def T(a):
    return torch.from_numpy(np.asarray(a))


def digest(params):
    h = hashlib.sha256()
    for k in params:
        h.update(k.encode())
        h.update(params[k].contiguous().numpy().tobytes())
    return h.hexdigest()


def test_param_inventory_xl():
    shapes = odit.param_shapes(odit.DiTConfig())
    assert len(shapes) == 6 + 28 * 6 + 4
    n = sum(int(np.prod(s)) for s in shapes.values())
    assert n == 264_657_040  # 28 x (7 h^2 + 7 h) + embeddings + final layer (attention-only DiT/XL, 16-ch latents, patch 1)


def test_timestep_features_layout():
    f = odit.timestep_features(torch.tensor([0, 3]), 256)
    assert torch.equal(f[0, :128], torch.ones(128)) and torch.equal(f[0, 128:], torch.zeros(128))  # [cos | sin]
    assert abs(float(f[1, 128]) - np.sin(3.0)) < 1e-6


def test_dit_tiny_forward():
    g = load("dit_tiny.npz")
    p = odit.seeded_params(TINY, 0)
    assert digest(p) == str(g["digest"])
    x, k = T(g["x"]), T(g["k"])
    np.testing.assert_allclose(odit.forward(p, TINY, x, k).numpy(), g["out"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(odit.forward(p, TINY, x[:, :3], k[:, :3]).numpy(), g["out_t3"], rtol=1e-4, atol=2e-5)


def test_dit_tiny_mlp_patch2():
    g = load("dit_tiny.npz")
    p = odit.seeded_params(TINY_MLP, 1)
    assert digest(p) == str(g["digest_mlp"])
    np.testing.assert_allclose(odit.forward(p, TINY_MLP, T(g["x_mlp"]), T(g["k"])).numpy(), g["out_mlp"], rtol=1e-4, atol=2e-5)


def test_dit_k600_forward():
    g = load("dit_k600.npz")
    cfg = odit.DiTConfig()
    p = odit.seeded_params(cfg, 0)
    assert digest(p) == str(g["digest"])
    taps = {}
    out = odit.forward(p, cfg, T(g["x"]), T(g["k"]), taps)
    for i in (0, 13, 27):
        np.testing.assert_allclose(float(taps[f"block{i}"].abs().mean()), float(g[f"block{i}_absmean"]), rtol=1e-4)
        np.testing.assert_allclose(taps[f"block{i}"][0, [0, 255, 700, 1279], :64].numpy(), g[f"block{i}_rows"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-3, atol=1e-3)


def test_discrete_cosine_tables_and_sampler():
    g = load("sampler_k600.npz")
    tb = sch.build_tables(beta_schedule="cosine")
    np.testing.assert_allclose(tb.alphas_cumprod.numpy(), g["alphas_cumprod"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(tb.sqrt_alphas_cumprod.numpy(), g["sqrt_alphas_cumprod"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(tb.sqrt_one_minus_alphas_cumprod.numpy(), g["sqrt_one_minus_alphas_cumprod"], rtol=1e-6, atol=1e-9)
    p = odit.seeded_params(SMALL, 2)
    assert digest(p) == str(g["digest"])
    model = lambda x, k, c, m: odit.forward(p, SMALL, x, k)
    noise = [T(g[f"noise{i}"]) for i in range(int(g["n_noise"]))]
    nfn = osm.replay_noise_fn(noise)
    cfg = osm.SamplerConfig(x_shape=(4, 16, 8), max_tokens=5, sampling_timesteps=4,
                            prediction_guidance=dict(name="vanilla", guidance_scale=2.0))
    diff = osm.Diffusion(tb, model, sampling_timesteps=4, is_continuous=False)
    out = osm.Sampler(cfg, diff, None, nfn).predict_videos(T(g["xs"]), 2, None)
    assert not nfn.queue
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-3, atol=2e-3)


# ---- DifferenceDiT3D, factorized matrix attention (the bash/k600 backbone) ----------------------------------------
DIFF_TINY = odit.DiffDiTConfig(hidden_size=128, depth=2, num_heads=4, in_channels=4, resolution=(16, 8), embed_col_dim=64, num_row_heads=4)
DIFF_TINY2 = odit.DiffDiTConfig(hidden_size=128, depth=1, num_heads=2, in_channels=4, resolution=(16, 8), embed_col_dim=64,
                                num_col_heads=2, num_row_heads=2, use_bias=False, mlp_ratio=0.0)
DIFF_WIDE = odit.DiffDiTConfig(depth=3)


def test_diffdit_param_inventory_k600():
    n = sum(int(np.prod(s)) for s in odit.diff_param_shapes(odit.DiffDiTConfig()).values())
    assert abs(n - 1358.2e6) < 0.1e6  # 1358.2 M parameters (probe of the reference module)


def test_diffdit_forward_vs_reference_fixture():
    g = load("diffdit.npz")
    x, k = T(g["x"]), T(g["k"])
    p1 = odit.diff_seeded_params(DIFF_TINY, 0)
    assert digest(p1) == str(g["digest"])
    np.testing.assert_allclose(odit.diff_forward(p1, DIFF_TINY, x, k).numpy(), g["out"], rtol=1e-4, atol=3e-5)
    np.testing.assert_allclose(odit.diff_forward(p1, DIFF_TINY, x[:, :6], k[:, :6]).numpy(), g["out_t6"], rtol=1e-4, atol=3e-5)
    p2 = odit.diff_seeded_params(DIFF_TINY2, 1)
    assert digest(p2) == str(g["digest2"])
    np.testing.assert_allclose(odit.diff_forward(p2, DIFF_TINY2, x, k).numpy(), g["out2"], rtol=1e-4, atol=3e-5)
    p3 = odit.diff_seeded_params(DIFF_WIDE, 2)
    assert digest(p3) == str(g["digestw"])
    np.testing.assert_allclose(odit.diff_forward(p3, DIFF_WIDE, T(g["xw"]), T(g["kw"])).numpy(), g["outw"], rtol=1e-3, atol=1e-3)


def test_difference_sampler_vs_reference_fixture():
    """DifferenceDFoTVideo = the DFoTVideo sampling path on the 2T interleaved (difference, frame) tokens."""
    g = load("sampler_k600_diff.npz")
    p = odit.diff_seeded_params(DIFF_TINY, 3)
    assert digest(p) == str(g["digest"])
    xs = T(g["xs"])
    diff = torch.diff(xs, dim=1, prepend=xs[:, :1])
    merged = torch.stack([diff, xs], dim=2).reshape(2, 10, 4, 16, 8)
    assert torch.equal(merged, T(g["merged"]))
    model = lambda x, k, c, m: odit.diff_forward(p, DIFF_TINY, x, k)
    noise = [T(g[f"noise{i}"]) for i in range(int(g["n_noise"]))]
    nfn = osm.replay_noise_fn(noise)
    cfg = osm.SamplerConfig(x_shape=(4, 16, 8), max_tokens=10, sampling_timesteps=3,
                            prediction_guidance=dict(name="vanilla", guidance_scale=1.5))
    diffusion = osm.Diffusion(sch.build_tables(beta_schedule="cosine"), model, sampling_timesteps=3, is_continuous=False)
    out = osm.Sampler(cfg, diffusion, None, nfn).predict_videos(merged, 4, None)
    assert not nfn.queue
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(out[:, 1::2].numpy(), g["gen"], rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(out[:, 0::2].numpy(), g["gen_diff"], rtol=1e-3, atol=2e-3)


def test_discrete_loss_weights_and_loss_vs_reference_fixture():
    g = load("discrete_loss.npz")
    tb = sch.build_tables(beta_schedule="cosine")
    k = T(g["k"])
    for key, kw in (("w_fused_096", dict(strategy="fused_min_snr", cum_snr_decay=0.96)),
                    ("w_fused_090", dict(strategy="fused_min_snr", cum_snr_decay=0.9)),
                    ("w_min_snr", dict(strategy="min_snr")), ("w_uniform", dict(strategy="uniform"))):
        np.testing.assert_allclose(osm.discrete_loss_weights(tb, k, **kw).numpy(), g[key], rtol=2e-5, atol=1e-9)
    p = odit.seeded_params(SMALL, 2)
    assert digest(p) == str(g["digest"])
    x_pred, loss = osm.discrete_training_loss(lambda x, kk, c, m: odit.forward(p, SMALL, x, kk), tb, T(g["x"]), k, T(g["noise"]),
                                              strategy="fused_min_snr", cum_snr_decay=0.96)
    np.testing.assert_allclose(x_pred.numpy(), g["x_pred"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=1e-3, atol=1e-5)


# ---- temporal / custom History Guidance (history sub-sequences, several gen segments) ----------------------------------
HG_TEMPORAL = {
    "temporal": dict(name="temporal", hist_subsequences=[[0], [1], [0, 1]], hist_weights=[0.5, 0.5, 1.0], gen_segments=[[0, 1], [1, 2]]),
    "custom": dict(name="custom", hist_segments=[dict(time_indices=[0, -1], freq_ranges=[[0.0, 1.0], [0.3, 1.0]],
                                                      freq_ranges_if_generated=[[0.1, 1.0]])],
                   hist_weights=[2.0], gen_segments=None),
}


def _small_diffusion(steps=50):
    p = odit.seeded_params(SMALL, 4)
    model = lambda x, k, c, m: odit.forward(p, SMALL, x, k)
    return p, osm.Diffusion(sch.build_tables(beta_schedule="cosine"), model, sampling_timesteps=steps, is_continuous=False)


def test_temporal_and_custom_guidance_step_vs_reference_fixture():
    from oracle import guidance as ohg
    g = load("hg_temporal.npz")
    p, diff = _small_diffusion()
    assert digest(p) == str(g["digest"])
    for sname, sc in HG_TEMPORAL.items():
        nfn = osm.replay_noise_fn([T(g[f"{sname}_noise{i}"]) for i in range(int(g[f"{sname}_n_noise"]))])
        gd = ohg.Guidance(ohg.make_scheme(**sc), T(g["cmask"]))
        assert gd.nfe == int(g[f"{sname}_nfe"])
        x_in, f_in, t_in, cm = gd.prepare(T(g["xs"]), T(g["frm"]), T(g["to"]), diff.q_sample, nfn)
        np.testing.assert_allclose(x_in.numpy(), g[f"{sname}_x_in"], rtol=1e-6, atol=1e-6)
        assert np.array_equal(f_in.numpy(), g[f"{sname}_from"]) and np.array_equal(t_in.numpy(), g[f"{sname}_to"])
        x_out = diff.ddim_step(x_in, f_in, t_in, None, cm, nfn("ddim", tuple(x_in.shape)))
        np.testing.assert_allclose(x_out.numpy(), g[f"{sname}_x_out"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(gd.compose(x_out).numpy(), g[f"{sname}_x_composed"], rtol=1e-4, atol=5e-4)
        assert not nfn.queue


def test_temporal_guidance_sampler_vs_reference_fixture():
    g = load("hg_temporal.npz")
    _, diff = _small_diffusion(steps=3)
    nfn = osm.replay_noise_fn([T(g[f"pred_noise{i}"]) for i in range(int(g["pred_n_noise"]))])
    cfg = osm.SamplerConfig(x_shape=(4, 16, 8), max_tokens=5, sampling_timesteps=3, prediction_guidance=HG_TEMPORAL["temporal"])
    out = osm.Sampler(cfg, diff, None, nfn).predict_videos(T(g["vid"]), 2, None)
    assert not nfn.queue
    np.testing.assert_allclose(out.numpy(), g["pred"], rtol=1e-3, atol=2e-3)


def test_refinement_sampler_vs_reference_fixture():
    """_sample_sequence_refine of the fork: DDIM steps + q_sample_from_x_k re-noising on the refinement ladder (RE10K schedule:
    with the K600 cosine schedule alphas_cumprod[T-1] is 0 in fp32 and the reference divides 0/0 for every context token)."""
    g = load("sampler_refine.npz")
    p = odit.seeded_params(SMALL, 2)
    assert digest(p) == str(g["digest"])
    tb = sch.build_tables()
    model = lambda x, k, c, m: odit.forward(p, SMALL, x, k)
    diff = osm.Diffusion(tb, model, sampling_timesteps=6, is_continuous=False)
    cfg = osm.SamplerConfig(x_shape=(4, 16, 8), max_tokens=5, sampling_timesteps=6)
    from oracle import guidance as hgo
    scheme = hgo.make_scheme(name="conditional")
    nfn = osm.replay_noise_fn([T(g[f"noise{i}"]) for i in range(int(g["n_noise"]))])
    out = osm.Sampler(cfg, diff, None, nfn).sample_sequence_refine(2, 2, 2, T(g["xs"]), T(g["mask"]), None, scheme)
    assert not nfn.queue
    assert np.isfinite(g["out"]).all()
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-3, atol=2e-3)
    # a padded window only ever re-noises, also on descending rows (scale > 1): NaN in the reference, and in its restatement
    assert bool(g["padded_window_is_nan"])
    nfn = osm.default_noise_fn(torch.Generator().manual_seed(0))
    out4 = osm.Sampler(cfg, diff, None, nfn).sample_sequence_refine(2, 2, 2, T(g["xs"])[:, :4], T(g["mask"])[:, :4], None, scheme, length=4)
    assert torch.isnan(out4).any()


def test_training_gradients_vs_reference_fixture():
    """the reference's own training loss (diffusion_model(xs, None, k) -> _reweight_loss with masks) differentiated by the reference:
    the oracle's restatement under autograd reproduces the loss, every gradient norm and the stored gradients (both K600 families)"""
    g = load("training_grads.npz")
    xs, k, masks = T(g["xs"]), T(g["k"]), T(g["masks"])
    tb = sch.build_tables(beta_schedule="cosine")
    merge = lambda a, b: torch.stack([a, b], dim=2).flatten(1, 2)
    for tag in ("dit", "diff"):
        if tag == "dit":
            params, fwd = odit.seeded_params(SMALL, 2), lambda ps, x, lv: odit.forward(ps, SMALL, x, lv)
            x_in, k_in, m_in = xs, k, masks
        else:
            params, fwd = odit.diff_seeded_params(DIFF_TINY, 3), lambda ps, x, lv: odit.diff_forward(ps, DIFF_TINY, x, lv)
            x_in, k_in, m_in = merge(torch.diff(xs, dim=1, prepend=xs[:, :1]), xs), merge(k, k), merge(masks, masks)
        assert digest(params) == str(g[f"{tag}_digest"])
        ps = {n: t.clone().requires_grad_() for n, t in params.items()}
        _, per_el = osm.discrete_training_loss(lambda x, lv, c, m: fwd(ps, x, lv), tb, x_in, k_in, T(g[f"{tag}_noise"]).clamp(-20, 20),
                                               strategy="fused_min_snr", cum_snr_decay=0.96)
        loss = (per_el * m_in[..., None, None, None]).mean()
        loss.backward()
        assert abs(loss.item() - float(g[f"{tag}_loss"])) < 1e-5 * abs(float(g[f"{tag}_loss"]))
        names = [str(n) for n in g[f"{tag}_names"]]
        assert names == list(ps)
        for n, ref_norm in zip(names, g[f"{tag}_norms"]):
            assert abs(float(ps[n].grad.norm()) - ref_norm) <= 2e-4 * ref_norm + 1e-9, (tag, n)
        stored = [key for key in g.files if key.startswith(f"{tag}_grad/")]
        assert len(stored) >= 8
        for key in stored:
            n = key.split("/", 1)[1]
            torch.testing.assert_close(ps[n].grad, T(g[key]), rtol=2e-3, atol=1e-6 + 2e-4 * float(T(g[key]).abs().max()))


def test_reconstruction_guidance_discrete_vs_reference_run():
    """cfg.diffusion.reconstruction_guidance > 0 on the DISCRETE cosine schedule (dfot_video.py:700-723, discrete_diffusion.py:485-513):
    the reference's own `DFoTVideo._predict_videos`, recorded with its draws (sampler_recon_k600.npz).  The first DDIM step is at the
    zero-terminal-SNR level, where the reference keeps the unguided x0 and still shifts the predicted noise -- the branch the
    continuous fixture never reaches.  Pins the oracle there; the guided run differs from the unguided one far beyond the tolerance."""
    g = load("sampler_recon_k600.npz")
    p = odit.seeded_params(SMALL, 2)
    assert digest(p) == str(g["digest"])
    tb = sch.build_tables(beta_schedule="cosine")
    outs = {}
    for tag, w in (("rg", float(g["weight"])), ("plain", 0.0)):
        noise = [T(g[f"{tag}_noise{i}"]) for i in range(int(g[f"{tag}_n_noise"]))]
        nfn = osm.replay_noise_fn(noise)
        cfg = osm.SamplerConfig(x_shape=(4, 16, 8), max_tokens=5, sampling_timesteps=3, prediction_guidance=dict(name="conditional"),
                                reconstruction_guidance=w)
        diff = osm.Diffusion(tb, lambda x, k, c, m: odit.forward(p, SMALL, x, k), sampling_timesteps=3, is_continuous=False)
        out = osm.Sampler(cfg, diff, None, nfn).predict_videos(T(g["xs"]), 2, None).detach()
        assert not nfn.queue
        np.testing.assert_allclose(out.numpy(), g[f"{tag}_out"], rtol=1e-3, atol=2e-3)
        outs[tag] = out
    moved = float((outs["rg"] - outs["plain"]).norm() / outs["plain"].norm())
    assert moved > 5e-3, moved
