"""Pins the CPU oracle against golden vectors captured from the reference's own source
(tools/make_golden.py).  fp32 tolerances: rtol 1e-5 / atol 1e-6 per op, 1e-4 whole backbone
(SURVEY.md section 8c)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import guidance as hg
from oracle import pose as opose
from oracle import sampler as osm
from oracle import schedule as sch
from oracle import uvit as ouvit

torch.set_num_threads(8)

TINY = ouvit.UViTConfig(channels=(32, 64, 72, 144), emb_channels=64, num_updown_blocks=(1, 1, 2),
                        num_mid_blocks=2, num_heads=9, resolution=16)
W64 = ouvit.UViTConfig(resolution=64)


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


def tiny_model(seed=0):
    p = ouvit.seeded_params(TINY, seed)
    return p, (lambda x, k, c, m: ouvit.forward(p, TINY, x, k, c, m))


def test_schedule_tables():
    g = load("schedule.npz")
    t = sch.build_tables()
    for name in ("alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"):
        np.testing.assert_allclose(getattr(t, name).numpy(), g[name], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(t.logsnr.numpy(), g["logsnr"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(sch.ddim_levels(1000, 50).numpy(), g["ddim_levels"])
    assert np.array_equal(sch.scheduling_matrix("full_sequence", 8, 0, 1000, 50).numpy(), g["sched_8_0"])
    assert np.array_equal(sch.scheduling_matrix("full_sequence", 5, 3, 1000, 50).numpy(), g["sched_5_3"])
    np.testing.assert_allclose(sch.training_logsnr(T(g["train_t"])).numpy(), g["train_logsnr"], rtol=1e-5, atol=1e-5)


def test_ray_encoding():
    g = load("ray_encoding.npz")
    poses = T(g["poses"])
    enc = opose.ray_encoding(poses, 8)
    # channel s of each 15-frequency block carries argument x*2^s*pi: one fp32 ulp of x moves the
    # top-frequency channel by ~2^14*pi*6e-8 = 3e-3, so tolerance scales with frequency
    np.testing.assert_allclose(enc.numpy(), g["enc8"], atol=2e-2)
    low = [c for c in range(180) if (c % 15) < 8]
    np.testing.assert_allclose(enc[:, :, low].numpy(), g["enc8"][:, :, low], atol=1e-4)
    enc256 = opose.ray_encoding(poses[:1], 256)[:, :, :, g["rows256"].tolist(), :]
    np.testing.assert_allclose(enc256[:, :, low].numpy(), g["enc256_rows"][:, :, low], atol=1e-4)
    np.testing.assert_allclose(enc256.numpy(), g["enc256_rows"], atol=2e-2)


def test_backbone_tiny():
    g = load("backbone_tiny.npz")
    p, model = tiny_model()
    assert digest(p) == str(g["digest"])
    cond = opose.ray_encoding(T(g["poses"]), 16)
    v = model(T(g["x"]), T(g["k"]), cond, T(g["mask"]))
    np.testing.assert_allclose(v.numpy(), g["v_masked"], rtol=1e-4, atol=1e-4)
    v = model(T(g["x"]), T(g["k"]), cond, None)
    np.testing.assert_allclose(v.numpy(), g["v_nomask"], rtol=1e-4, atol=1e-4)


def test_backbone_re10k_widths():
    g = load("backbone_w64.npz")
    p = ouvit.seeded_params(W64, 3)
    assert digest(p) == str(g["digest"])
    cond = opose.ray_encoding(T(g["poses"]), 64)
    v = ouvit.forward(p, W64, T(g["x"]), T(g["k"]), cond, T(g["mask"]))
    ref = T(g["v"])
    rel = ((v - ref).norm() / ref.norm()).item()
    assert rel < 1e-4, rel
    np.testing.assert_allclose(v.numpy(), g["v"], rtol=1e-3, atol=2e-4)


SCHEMES = {
    "conditional": dict(name="conditional"),
    "vanilla": dict(name="vanilla", guidance_scale=4.0),
    "stabilized_vanilla": dict(name="stabilized_vanilla", guidance_scale=4.0, stabilization_level=0.02),
    "fractional": dict(name="fractional", guidance_scale=3.0, freq_scale=0.4),
}


@pytest.mark.parametrize("sname", list(SCHEMES))
def test_step_trace(sname):
    g = load("step_trace.npz")
    p, model = tiny_model()
    assert digest(p) == str(g["digest"])
    diff = osm.Diffusion(sch.build_tables(), model)
    noise = [T(g[f"{sname}_noise{i}"]) for i in range(int(g[f"{sname}_n_noise"]))]
    nfn = osm.replay_noise_fn(noise)
    scheme = hg.make_scheme(**SCHEMES[sname])
    gd = hg.Guidance(scheme, T(g["cmask"]))
    assert gd.nfe == int(g[f"{sname}_nfe"])
    x_in, f_in, t_in, cm = gd.prepare(T(g["xs"]), T(g["frm"]), T(g["to"]), diff.q_sample, nfn)
    np.testing.assert_allclose(x_in.numpy(), g[f"{sname}_x_in"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(f_in.numpy(), g[f"{sname}_from"])
    assert np.array_equal(t_in.numpy(), g[f"{sname}_to"])
    if cm is None:
        assert g[f"{sname}_cond_mask"].size == 0
    else:
        assert np.array_equal(cm.numpy(), g[f"{sname}_cond_mask"])
    cond = opose.ray_encoding(T(g["conds"]).repeat_interleave(gd.nfe, 0), 16)
    x_out = diff.ddim_step(x_in, f_in, t_in, cond, cm, nfn("ddim", tuple(x_in.shape)))
    np.testing.assert_allclose(x_out.numpy(), g[f"{sname}_x_out"], rtol=1e-4, atol=1e-4)
    xc = gd.compose(x_out)
    np.testing.assert_allclose(xc.numpy(), g[f"{sname}_x_composed"], rtol=1e-4, atol=5e-4)
    assert not nfn.queue


def _sampler(cfg, model, nfn):
    diff = osm.Diffusion(sch.build_tables(), model, sampling_timesteps=cfg.sampling_timesteps)
    return osm.Sampler(cfg, diff, lambda c: opose.ray_encoding(c, cfg.x_shape[-1]), nfn)


def test_sampler_8_frames():
    g = load("sampler_8f.npz")
    p, model = tiny_model()
    assert digest(p) == str(g["digest"])
    noise = [T(g[f"noise{i}"]) for i in range(int(g["n_noise"]))]
    nfn = osm.replay_noise_fn(noise)
    cfg = osm.SamplerConfig(x_shape=(3, 16, 16), sampling_timesteps=3,
                            prediction_guidance=dict(name="vanilla", guidance_scale=4.0))
    out = _sampler(cfg, model, nfn).predict_videos(T(g["xs"]), 1, T(g["conds"]))
    assert not nfn.queue
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-3, atol=2e-3)


def test_sampler_200_frames_plan_and_result():
    g = load("sampler_200f.npz")
    p, model = tiny_model()
    assert digest(p) == str(g["digest"])
    init_gen = torch.Generator().manual_seed(0)
    torch.manual_seed(777)
    shapes = []

    def nfn(tag, shape):
        shapes.append(list(shape) + [0] * (6 - len(shape)))
        t = torch.randn(shape, generator=init_gen) if tag == "init" else torch.randn(shape)
        return t if tag == "excluded" else t.clamp(-20, 20)

    cfg = osm.SamplerConfig(
        x_shape=(3, 16, 16), sampling_timesteps=2, keyframe_density=0.0625,
        prediction_guidance=dict(name="stabilized_vanilla", guidance_scale=4.0, stabilization_level=0.02),
        interpolation_guidance=dict(name="vanilla", guidance_scale=1.5), interpolation_max_batch_size=4)
    s = _sampler(cfg, model, nfn)
    out = s.predict_videos(T(g["xs"]), 1, T(g["conds"]))
    assert [t["batch"] for t in s.trace] == g["call_batches"].tolist()
    masks = np.concatenate([t["context_mask"].numpy() for t in s.trace], 0)
    assert np.array_equal(masks, g["call_masks"])
    assert np.array_equal(np.array(shapes), g["draw_shapes"])
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-3, atol=2e-3)


def test_training_loss():
    g = load("training_loss.npz")
    p, model = tiny_model()
    cond = opose.ray_encoding(T(g["poses"]), 16)
    x_pred, loss = osm.training_loss(model, T(g["x"]), cond, T(g["t"]), T(g["noise"]).clamp(-20, 20))
    np.testing.assert_allclose(x_pred.numpy(), g["x_pred"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("kind", ["interleaved", "gibbs", "autoregressive"])
def test_other_scheduling_matrices(kind):
    g = load("schedule_extra.npz")
    s = int(g["sampling_steps"])
    for h, p in ((8, 0), (5, 3)):
        assert torch.equal(sch.scheduling_matrix(kind, h, p, 1000, s), T(g[f"{kind}_{h}_{p}"]))


TRAIN = ouvit.UViTConfig(channels=(128, 128, 128, 256), emb_channels=128, num_updown_blocks=(1, 1, 1), num_mid_blocks=1, num_heads=2, resolution=128)


def test_uvit_training_gradients_vs_reference_fixture():
    """the reference's own RE10K-style training loss (ContinuousDiffusion.forward through UViT3DPose, _reweight_loss with masks) differentiated
    by the reference: the oracle's restatement under autograd reproduces the loss, every gradient norm and the stored gradients"""
    g = load("training_grads_uvit.npz")
    params = ouvit.seeded_params(TRAIN, 6)
    assert digest(params) == str(g["digest"])
    ps = {n: v.clone().requires_grad_(not n.endswith(("freqs", "phases"))) for n, v in params.items()}
    cond = opose.ray_encoding(T(g["poses"]), 128)
    _, per_el = osm.training_loss(lambda x, k, c, m: ouvit.forward(ps, TRAIN, x, k, c, m), T(g["xs"]), cond, T(g["k"]), T(g["noise"]).clamp(-20, 20))
    loss = (per_el * T(g["masks"])[..., None, None, None]).mean()
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    names = [str(n) for n in g["names"]]  # the reference module's own parameter order (up blocks are registered before mid blocks)
    assert sorted(names) == sorted(n for n in ps if ps[n].requires_grad)
    for n, ref_norm in zip(names, g["norms"]):
        assert abs(float(ps[n].grad.norm()) - ref_norm) <= 1e-3 * ref_norm + 1e-9, n
    stored = [key for key in g.files if key.startswith("grad/")]
    assert len(stored) >= 20
    for key in stored:
        ref = T(g[key])
        torch.testing.assert_close(ps[key[5:]].grad, ref, rtol=5e-3, atol=1e-7 + 1e-3 * float(ref.abs().max()))


@pytest.mark.parametrize("tag", ["eta", "ddpm"])
def test_stochastic_sampling_steps_vs_reference_runs(tag):
    """eta > 0 DDIM and DDPM sampling (discrete_diffusion.py:423-452, 515-538): the reference's own `_predict_videos` runs with every
    normal draw recorded; the oracle replays the draws and must reproduce the result"""
    g = load("sampler_stochastic.npz")
    p, model = tiny_model()
    assert digest(p) == str(g["digest"])
    noise = [T(g[f"{tag}_noise{i}"]) for i in range(int(g[f"{tag}_n_noise"]))]
    nfn = osm.replay_noise_fn(noise)
    if tag == "eta":
        tables, steps, ts, eta = sch.build_tables(), 3, 1000, 0.5
    else:
        tables, steps, ts, eta = sch.build_tables(timesteps=6), 6, 6, 0.0
        np.testing.assert_allclose(tables.alphas_cumprod.numpy(), g["ddpm_alphas_cumprod"], rtol=1e-6)
        np.testing.assert_allclose(tables.posterior_mean_coef1.numpy(), g["ddpm_coef1"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(tables.posterior_mean_coef2.numpy(), g["ddpm_coef2"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(tables.posterior_log_variance_clipped.numpy(), g["ddpm_log_var"], rtol=1e-5, atol=1e-6)
    cfg = osm.SamplerConfig(x_shape=(3, 16, 16), timesteps=ts, sampling_timesteps=steps,
                            prediction_guidance=dict(name="vanilla", guidance_scale=4.0))
    diff = osm.Diffusion(tables, model, sampling_timesteps=steps, eta=eta)
    s = osm.Sampler(cfg, diff, lambda c: opose.ray_encoding(c, 16), nfn)
    out = s.predict_videos(T(g[f"{tag}_xs"]), 1, T(g[f"{tag}_conds"]))
    assert not nfn.queue
    np.testing.assert_allclose(out.numpy(), g[f"{tag}_out"], rtol=1e-3, atol=2e-3)


def test_reconstruction_guidance_vs_reference_run():
    """cfg.diffusion.reconstruction_guidance > 0 (dfot_video.py:700-723, discrete_diffusion.py:485-513): the reference's own
    `_predict_videos` run of the tiny pose model (conditional guidance, 2 context frames, 3 DDIM steps, weight 400) with every normal
    draw recorded; the oracle (autograd through oracle.uvit) replays the draws and must reproduce it -- and the unguided run, which
    the guided one must differ from by far more than the tolerance"""
    g = load("sampler_recon.npz")
    p, model = tiny_model()
    assert digest(p) == str(g["digest"])
    outs = {}
    for tag, w in (("rg", float(g["weight"])), ("plain", 0.0)):
        noise = [T(g[f"{tag}_noise{i}"]) for i in range(int(g[f"{tag}_n_noise"]))]
        nfn = osm.replay_noise_fn(noise)
        cfg = osm.SamplerConfig(x_shape=(3, 16, 16), sampling_timesteps=3, prediction_guidance=dict(name="conditional"),
                                reconstruction_guidance=w)
        s = osm.Sampler(cfg, osm.Diffusion(sch.build_tables(), model, sampling_timesteps=3), lambda c: opose.ray_encoding(c, 16), nfn)
        outs[tag] = s.predict_videos(T(g["xs"]), 2, T(g["conds"])).detach()
        assert not nfn.queue
        np.testing.assert_allclose(outs[tag].numpy(), g[f"{tag}_out"], rtol=1e-3, atol=2e-3)
    assert np.abs(g["rg_out"] - g["plain_out"]).max() > 0.05
