"""CPU tests of the product's host-side planning (no device work): schedule tables, scheduling matrices and the
History-Guidance branch planner, checked against the golden vectors captured from the reference."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from dfot_amd import DiffusionConfig, HistoryGuidance, Schedule


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def test_schedule_tables_match_reference_buffers():
    g = load("schedule.npz")
    s = Schedule(DiffusionConfig())
    np.testing.assert_allclose(s.alphas_cumprod, g["alphas_cumprod"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(s.sqrt_alphas_cumprod, g["sqrt_alphas_cumprod"], rtol=1e-6)
    np.testing.assert_allclose(s.sqrt_one_minus_alphas_cumprod, g["sqrt_one_minus_alphas_cumprod"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(s.logsnr, g["logsnr"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(s.ddim_idx_to_noise_level(np.arange(51)), g["ddim_levels"])
    assert np.array_equal(s.scheduling_matrix("full_sequence", 8, 0), g["sched_8_0"])
    assert np.array_equal(s.scheduling_matrix("full_sequence", 5, 3), g["sched_5_3"])


SCHEMES = {
    "conditional": dict(name="conditional"),
    "vanilla": dict(name="vanilla", guidance_scale=4.0),
    "stabilized_vanilla": dict(name="stabilized_vanilla", guidance_scale=4.0, stabilization_level=0.02),
    "fractional": dict(name="fractional", guidance_scale=3.0, freq_scale=0.4),
}


@pytest.mark.parametrize("sname", list(SCHEMES))
def test_branch_plan_matches_reference_prepare(sname):
    """levels / cond-mask of every branch and the q_sample coefficients reproduce the reference's prepare()."""
    g = load("step_trace.npz")
    hg = HistoryGuidance.from_config(dict(SCHEMES[sname], visualize=False), timesteps=1000)
    plan = hg.plan(g["cmask"], g["frm"], g["to"])
    assert plan.nfe == int(g[f"{sname}_nfe"])
    b, t = g["cmask"].shape
    assert np.array_equal(plan.levels.reshape(b * plan.nfe, t), g[f"{sname}_from"])
    assert np.array_equal(plan.to_levels.reshape(b * plan.nfe, t), g[f"{sname}_to"])
    cm = g[f"{sname}_cond_mask"]
    if cm.size == 0:
        assert plan.cond_masked is None
    else:
        assert np.array_equal(np.tile(plan.cond_masked, b), cm)
    # x_in = qa * x + qb * noise must reproduce the reference's prepared tensor
    s = Schedule(DiffusionConfig())
    lv = plan.levels.reshape(b * plan.nfe, t)
    repl = plan.replace.reshape(b * plan.nfe, t)
    qa_c, qb_c = s.q_sample_coef(lv)
    qa = np.where(repl, qa_c, 1.0)[..., None, None, None]
    qb = np.where(repl, qb_c, 0.0)[..., None, None, None]
    x = np.repeat(g["xs"], plan.nfe, axis=0)
    if repl.any():
        n0 = g[f"{sname}_noise0"]
        if hg.is_simple:  # the reference draws (B,T,...) for the unconditional branch only
            noise = np.zeros_like(x).reshape(b, plan.nfe, *x.shape[1:])
            noise[:, 0] = n0
            noise = noise.reshape(x.shape)
        else:
            noise = n0
    else:
        noise = np.zeros_like(x)
    np.testing.assert_allclose(qa * x + qb * noise, g[f"{sname}_x_in"], rtol=1e-6, atol=1e-6)
    # composition weights reproduce compose(): sum_h w_h x_h on generated tokens
    xo = g[f"{sname}_x_out"].reshape(b, plan.nfe, *x.shape[1:])
    comp = (xo * plan.weights.reshape(1, -1, 1, 1, 1, 1)).sum(1)
    np.testing.assert_allclose(comp, g[f"{sname}_x_composed"], rtol=1e-4, atol=1e-4)


def test_ddim_coefficients_reproduce_reference_step():
    g = load("step_trace.npz")
    s = Schedule(DiffusionConfig())
    frm, to = g["conditional_from"], g["conditional_to"]
    sa, s1, an, cn, keep, sigma = s.ddim_coef(frm, to)
    assert (sigma == 0).all()
    assert np.array_equal(keep.astype(bool), frm == to)
    # clean tokens (level -1): an = 1, cn = 0
    assert np.allclose(an[to < 0], 1.0) and np.allclose(cn[to < 0], 0.0)


def _random_poses(b, t, seed):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    rot = Rotation.random(b * t, random_state=seed).as_matrix().reshape(b, t, 3, 3)
    rt = np.concatenate([rot, rng.normal(size=(b, t, 3, 1))], -1).reshape(b, t, 12)
    return np.concatenate([np.tile([0.5, 0.9, 0.5, 0.5], (b, t, 1)), rt], -1).astype(np.float32)


def test_pose_options_host_algebra():
    """normalize_by mean / bound / interpolation of masked poses (the reference does these with roma, absent here: parity unpinned
    against it).  Checked against SciPy's Rotation / Slerp -- the construction roma documents -- and against the oracle."""
    import torch
    from scipy.spatial.transform import Rotation, Slerp
    from dfot_amd import pose
    from oracle import pose as opose
    raw = _random_poses(2, 8, 3)
    rot = raw[..., 4:].reshape(2, 8, 3, 4)[..., :3]
    q = pose._to_quat(rot)
    qs = Rotation.from_matrix(rot.reshape(-1, 3, 3)).as_quat().reshape(2, 8, 4)
    assert np.minimum(np.abs(q - qs).max(-1), np.abs(q + qs).max(-1)).max() < 1e-6
    np.testing.assert_allclose(pose._to_rotmat(q), rot, atol=1e-6)
    # interpolation: frames 2..4 and 6 masked -> slerp/lerp between 1 and 5, 5 and 7; frame 0 masked -> held from frame 1
    mask = np.zeros((2, 8), bool)
    mask[0, [0, 2, 3, 4, 6]] = True
    r2, t2 = pose.interpolate_masked(rot, raw[..., 4:].reshape(2, 8, 3, 4)[..., 3], mask)
    np.testing.assert_allclose(r2[1], rot[1], atol=1e-6)  # unmasked video: only the quaternion round trip
    np.testing.assert_allclose(r2[0, 0], rot[0, 1], atol=1e-6)
    sl = Slerp([1, 5], Rotation.from_matrix(rot[0, [1, 5]]))
    np.testing.assert_allclose(r2[0, 1:6], sl([1, 2, 3, 4, 5]).as_matrix(), atol=2e-6)
    tr = raw[0, :, 4:].reshape(8, 3, 4)[..., 3]
    np.testing.assert_allclose(t2[0, 3], 0.5 * (tr[1] + tr[5]), atol=1e-6)
    np.testing.assert_allclose(t2[0, 6], 0.5 * (tr[5] + tr[7]), atol=1e-6)
    # all options against the oracle's restatement
    for kw in (dict(normalize_by="mean"), dict(normalize_by="first", bound=1.0), dict(normalize_by="mean", bound=0.5, interpolate_mask=mask)):
        world = pose.normalize_poses(raw, **kw)
        okw = dict(kw)
        if "interpolate_mask" in okw:
            okw["interpolate_mask"] = torch.from_numpy(mask)
        _, r_ref, t_ref = opose.normalized_poses(torch.from_numpy(raw), **okw)
        rt = world[..., 4:].reshape(2, 8, 3, 4)
        np.testing.assert_allclose(rt[..., :3], r_ref.numpy(), atol=2e-6)
        np.testing.assert_allclose(rt[..., 3], t_ref.numpy(), atol=5e-6)
        assert np.array_equal(world[..., :4], raw[..., :4])
    # normalised by the first frame: frame 0 becomes the identity pose
    w = pose.normalize_poses(raw, "first")
    np.testing.assert_allclose(w[:, 0, 4:].reshape(2, 3, 4), np.tile(np.eye(3, 4, dtype=np.float32), (2, 1, 1)), atol=1e-6)
    # bound: every axis of the camera positions fits [-bound, bound] and touches it
    w = pose.normalize_poses(raw, "first", bound=0.25)
    assert np.allclose(np.abs(w[..., 4:].reshape(2, 8, 3, 4)[..., 3]).max(axis=1), 0.25, atol=1e-6)
    with pytest.raises(ValueError):
        pose.normalize_poses(raw, "median")


@pytest.mark.parametrize("kind", ["interleaved", "gibbs", "autoregressive"])
def test_other_scheduling_matrices(kind):
    g = np.load(os.path.join(GOLDEN, "schedule_extra.npz"))
    s = Schedule(DiffusionConfig(sampling_timesteps=int(g["sampling_steps"])))
    for h, p in ((8, 0), (5, 3)):
        assert np.array_equal(s.scheduling_matrix(kind, h, p), g[f"{kind}_{h}_{p}"])


def test_discrete_cosine_schedule_matches_reference_buffers():
    g = np.load(os.path.join(GOLDEN, "sampler_k600.npz"))
    s = Schedule(DiffusionConfig(beta_schedule="cosine", is_continuous=False))
    np.testing.assert_allclose(s.alphas_cumprod, g["alphas_cumprod"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(s.sqrt_one_minus_alphas_cumprod, g["sqrt_one_minus_alphas_cumprod"], rtol=1e-6, atol=1e-9)
    k = np.array([[-1, 0, 17, 999]])
    assert np.array_equal(s.model_level(k), np.array([[0, 0, 17, 999]], np.float32))  # clamped level index, exact in fp32


def test_discrete_loss_weights_match_reference():
    g = np.load(os.path.join(GOLDEN, "discrete_loss.npz"))
    s = Schedule(DiffusionConfig(beta_schedule="cosine", is_continuous=False))
    k = g["k"]
    # fp32 on both sides with a different operation order: 1 - keep * (1 - snr/clip) cancels for snr -> 0 (abs error ~2e-7)
    np.testing.assert_allclose(s.loss_weights(k, "fused_min_snr", cum_snr_decay=0.96), g["w_fused_096"], rtol=1e-4, atol=5e-7)
    np.testing.assert_allclose(s.loss_weights(k, "fused_min_snr", cum_snr_decay=0.9), g["w_fused_090"], rtol=1e-4, atol=5e-7)
    np.testing.assert_allclose(s.loss_weights(k, "min_snr"), g["w_min_snr"], rtol=1e-4, atol=5e-7)
    np.testing.assert_allclose(s.loss_weights(k, "uniform"), g["w_uniform"])


def test_temporal_and_custom_guidance_plans_match_reference():
    """History sub-sequences (time_indices) and several gen segments: branch levels, excluded tokens and per-token weights."""
    g = np.load(os.path.join(GOLDEN, "hg_temporal.npz"))
    schemes = {
        "temporal": dict(name="temporal", hist_subsequences=[[0], [1], [0, 1]], hist_weights=[0.5, 0.5, 1.0], gen_segments=[[0, 1], [1, 2]]),
        "custom": dict(name="custom", hist_segments=[dict(time_indices=[0, -1], freq_ranges=[[0.0, 1.0], [0.3, 1.0]],
                                                          freq_ranges_if_generated=[[0.1, 1.0]])], hist_weights=[2.0]),
    }
    for sname, sc in schemes.items():
        hg = HistoryGuidance.from_config(sc, timesteps=1000)
        assert not hg.is_simple
        plan = hg.plan(g["cmask"], g["frm"], g["to"])
        assert plan.nfe == int(g[f"{sname}_nfe"])
        assert np.array_equal(plan.levels.reshape(-1, 5), g[f"{sname}_from"])
        assert np.array_equal(plan.to_levels.reshape(-1, 5), g[f"{sname}_to"])
    plan = HistoryGuidance.from_config(schemes["temporal"], timesteps=1000).plan(g["cmask"], g["frm"], g["to"])
    assert plan.n_gen == 2 and plan.tok_weights.shape == (plan.nfe, 5)
    # gen token 1 (sequence position 3) is covered by both segments -> its weights are halved; excluded tokens weigh 0
    w = plan.tok_weights.reshape(-1, 2, 5)
    assert np.allclose(w[:, 0, 4], 0) and np.allclose(w[:, 1, 2], 0)
    assert np.allclose(w[:, 0, 3], 0.5 * w[:, 0, 2]) and np.allclose(w[:, 1, 3], 0.5 * w[:, 1, 4])
    assert plan.excluded[0, 0].tolist() == [False, False, False, False, True]
    assert plan.excluded[0, 1].tolist() == [False, False, True, False, False]


def test_training_noise_levels_match_reference():
    """_get_training_noise_levels with a seeded CPU generator: identical draws, context handling and loss masks."""
    import torch
    from dfot_amd import ContextTraining, TrainingNoise
    g = np.load(os.path.join(GOLDEN, "training_noise.npz"))
    cases = {
        "indep": dict(noise_level="random_independent"),
        "uniform": dict(noise_level="random_uniform"),
        "interleaved": dict(noise_level="interleaved"),
        "ufuture": dict(noise_level="random_independent", uniform_future=True),
        "fixed": dict(noise_level="random_independent", fixed_context=ContextTraining(enabled=True, dropout=0.5)),
        "variable": dict(noise_level="random_uniform", variable_context=ContextTraining(enabled=True, prob=0.25, dropout=0.3)),
    }
    for tag, cont, nt, mkey in (("c", True, 8, "masks8"), ("d", False, 5, "masks5")):
        for name, kw in cases.items():
            tn = TrainingNoise(is_continuous=cont, n_context_tokens=int(g[f"{tag}_n_context"]), **kw)
            lv, mk = tn.sample(3, nt, torch.from_numpy(g[mkey]), torch.Generator().manual_seed(123), training=True)
            assert np.array_equal(lv.numpy(), g[f"{tag}_{name}_levels"]), (tag, name)
            assert np.array_equal(mk.numpy(), g[f"{tag}_{name}_masks"]), (tag, name)


def test_refine_scheduling_matrix_matches_reference():
    g = np.load(os.path.join(GOLDEN, "schedule_extra.npz"))
    s = Schedule(DiffusionConfig(sampling_timesteps=50))
    assert np.array_equal(s.refine_scheduling_matrix(5, goback_length=20, n_goback=2, padding=3), g["refine50_5_3"])


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 3: the PRODUCT planner (sliding key-frame windows + interpolation stages + batching) against the plan
# recorded from the reference's own `_predict_videos` run (dfot_video.py:114-179,181-360,362-514)
# ---------------------------------------------------------------------------------------------------------------------
class _RecordingNoise:
    """strict-order noise source: the sampler consumes every draw the reference makes; shapes are logged like the fixture"""
    strict_order = True

    def __init__(self):
        self.shapes = []

    def __call__(self, tag, shape):
        import torch
        self.shapes.append(list(shape) + [0] * (6 - len(shape)))
        return torch.zeros(shape)


def _dry_sampler(noise_fn, **kw):
    import dfot_amd
    cfg = dfot_amd.SamplerConfig(
        x_shape=(3, 16, 16), diffusion=DiffusionConfig(sampling_timesteps=2), keyframe_density=0.0625,
        prediction_guidance=dict(name="stabilized_vanilla", guidance_scale=4.0, stabilization_level=0.02),
        interpolation_guidance=dict(name="vanilla", guidance_scale=1.5), interpolation_max_batch_size=4, **kw)
    s = dfot_amd.DFoTVideoPoseSampler(cfg, backbone=None, noise_fn=noise_fn)
    s.device, s.dry_run = "cpu", True   # plan everything, launch nothing
    return s


def test_product_planner_200_frames_equals_reference_plan():
    import torch
    g = load("sampler_200f.npz")
    rec = _RecordingNoise()
    s = _dry_sampler(rec)
    out = s._predict_videos(torch.from_numpy(g["xs"]), 1, torch.from_numpy(g["conds"]))
    assert out.shape == g["out"].shape
    # 14 sampler calls: 2 sequential key-frame windows, then 11 windows (3 batches) and 35 windows (9 batches)
    assert [t["batch"] for t in s.trace] == g["call_batches"].tolist()
    masks = np.concatenate([t["context_mask"] for t in s.trace], 0)
    assert np.array_equal(masks, g["call_masks"])
    # every noise draw of the reference run, in order, with its shape
    assert np.array_equal(np.array(rec.shapes), g["draw_shapes"])
    # window-forwards: 48 windows x NFE 2 x 2 DDIM steps (96 per step)
    assert s.window_forwards == 96 * 2


def test_interpolation_plan_windows_for_200_frames():
    """window index sets of the two plan stages: stage 1 = Case-1 windows (8 frames spread by linspace between two key frames,
    mask [1,0x6,1]), stage 2 = chunk windows over the remaining gaps (dfot_video.py:219-261)"""
    import torch
    s = _dry_sampler(_RecordingNoise())
    n = 200
    keys = torch.linspace(0, n - 1, round(0.0625 * n)).round().long()
    known = np.zeros(n, bool)
    known[keys.numpy()] = True
    plan = s._interpolation_plan(known)
    assert [len(st) for st in plan] == [11, 35]
    for w, (l, r) in zip(plan[0], zip(keys[:-1].tolist(), keys[1:].tolist())):
        assert np.array_equal(w, torch.linspace(l, r, 8).round().long().numpy())
    covered = known.copy()
    for st in plan:
        for w in st:
            assert len(w) <= 8 and np.all(np.diff(w) > 0)
            covered[w] = True
    assert covered.all()


def test_keyframe_noise_keys_do_not_depend_on_the_previous_batch_or_rank(monkeypatch):
    """with WindowKeyedNoise the replicated key-frame windows must draw the same noise on every rank and on every
    call of one sampler (their key is the sliding-window index, not what the last interpolation batch left behind)."""
    import torch
    from dfot_amd import parallel
    g = load("sampler_200f.npz")
    xs, conds = torch.from_numpy(g["xs"]), torch.from_numpy(g["conds"])

    def run(world, rank):
        log = []

        class Logged(parallel.WindowKeyedNoise):
            def __call__(self, tag, shape):
                t = super().__call__(tag, shape)
                if max(self.keys) < 100000:  # key-frame windows
                    log.append((self.keys, tag, tuple(shape), float(t.double().sum())))
                return t

        s = _dry_sampler(Logged(7, device="cpu"))
        s.shard_windows = world > 1
        monkeypatch.setattr(parallel, "world_info", lambda group=None: (world, rank))
        monkeypatch.setattr(parallel, "gather_windows",
                            lambda local, n, group=None: local.new_zeros((n, *local.shape[1:])))
        s._predict_videos(xs, 1, conds)
        first = list(log)
        del log[:]
        s._predict_videos(xs, 1, conds)   # second call on the SAME sampler (what bench.py --workload 200f does)
        return first, list(log)

    single = run(1, 0)
    r0, r1 = run(2, 0), run(2, 1)
    assert len(single[0]) > 0
    for call in (0, 1):
        assert r0[call] == r1[call] == single[call]
    # a draw whose rows cannot be split over the current keys is an error, not a silent re-use of stale keys
    nz = parallel.WindowKeyedNoise(1, device="cpu")
    nz.set_windows([1, 2, 3])
    with pytest.raises(ValueError):
        nz("init", (4, 8, 3, 4, 4))


def test_host_tensors_are_refused_before_any_launch():
    """the Python boundary never hands a host pointer (or a strided view / wrong dtype) to a kernel"""
    import torch
    from dfot_amd import capi
    with pytest.raises(ValueError, match="GPU memory only"):
        capi.ptr(torch.zeros(4), name="x")
    assert capi.ptr(None).value in (None, 0)
    with pytest.raises(ValueError, match="noise_levels is on cpu"):
        capi.require_device(torch.device("cuda", 0), noise_levels=torch.zeros(2, 8))
    x = torch.zeros(2, 8, 3, 4, 4)
    with pytest.raises(ValueError):
        torch.ops.dfot.hg_prepare(x, None, torch.zeros(4, 8), torch.zeros(4, 8), 2)
    with pytest.raises(ValueError):   # wrong table shape is caught before the device check
        torch.ops.dfot.hg_prepare(x, None, torch.zeros(2, 8), torch.zeros(4, 8), 2)
    # row-strided matrices (column blocks of wider ones) go through ptr_rows, which refuses host memory and anything else strided
    with pytest.raises(ValueError, match="GPU memory only"):
        capi.ptr_rows(torch.zeros(4, 8)[:, 2:6], name="dfilm")
    assert capi.ptr_rows(None).value in (None, 0)


@pytest.mark.parametrize("tag", ["eta", "ddpm"])
def test_stochastic_sampling_plan_vs_reference_run(tag):
    """eta > 0 DDIM / DDPM (discrete_diffusion.py:423-452,515-538): the product's schedule tables equal the reference's buffers and
    the planner consumes exactly the reference's normal draws (count, order, shapes) -- golden recorded from the reference run"""
    import torch
    import dfot_amd
    g = load("sampler_stochastic.npz")
    dcfg = DiffusionConfig(sampling_timesteps=3, ddim_sampling_eta=0.5) if tag == "eta" else DiffusionConfig(timesteps=6, sampling_timesteps=6)
    s = Schedule(dcfg)
    assert s.is_ddim_sampling == (tag == "eta")
    if tag == "ddpm":
        np.testing.assert_allclose(s.alphas_cumprod, g["ddpm_alphas_cumprod"], rtol=1e-6)
        np.testing.assert_allclose(s.posterior_mean_coef1, g["ddpm_coef1"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(s.posterior_mean_coef2, g["ddpm_coef2"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(s.posterior_log_variance_clipped, g["ddpm_log_var"], rtol=1e-5, atol=1e-6)
        # the fused-step form of the posterior mean: (c1 + c2*sa) x0 + (c2*s1) eps == c1 x0 + c2 x with x = sa x0 + s1 eps
        k = np.arange(6)[None]
        sa, s1, an, cn, keep, sigma = s.ddpm_coef(k)
        x0, eps = 0.7, -1.3
        np.testing.assert_allclose(an * x0 + cn * eps, s.posterior_mean_coef1[k] * x0 + s.posterior_mean_coef2[k] * (sa * x0 + s1 * eps), rtol=1e-5)
        assert sigma[0, 0] == 0 and (sigma[0, 1:] > 0).all() and not keep.any()
    rec = _RecordingNoise()
    cfg = dfot_amd.SamplerConfig(x_shape=(3, 16, 16), diffusion=dcfg, prediction_guidance=dict(name="vanilla", guidance_scale=4.0))
    smp = dfot_amd.DFoTVideoPoseSampler(cfg, backbone=None, noise_fn=rec)
    smp.device, smp.dry_run = "cpu", True
    smp._predict_videos(torch.from_numpy(g[f"{tag}_xs"]), 1, torch.from_numpy(g[f"{tag}_conds"]))
    ref_shapes = [list(g[f"{tag}_noise{i}"].shape) for i in range(int(g[f"{tag}_n_noise"]))]
    assert [sh[:len(r)] for sh, r in zip(rec.shapes, ref_shapes)] == ref_shapes and len(rec.shapes) == len(ref_shapes)


def test_training_schedule_comes_from_the_config_and_lr_warmup():
    """the training / validation loss reads logsnr limits, shift, loss-weight bias from DiffusionConfig (and refuses what the
    engine does not implement); the reference's constant_with_warmup lr schedule"""
    import torch
    from dfot_amd.training import lr_at_step
    t = torch.linspace(0, 1, 9).view(1, 9)
    base = DiffusionConfig().training_logsnr_tables(t)
    g = load("schedule.npz")   # train_logsnr: the reference's training_schedule at 33 points of [0, 1]
    ref = DiffusionConfig().training_logsnr_tables(torch.from_numpy(g["train_t"]))[0].numpy()
    np.testing.assert_allclose(ref, g["train_logsnr"], rtol=1e-5, atol=1e-5)
    other = DiffusionConfig(logsnr_min=-10.0, logsnr_max=12.0, training_schedule_shift=0.5, loss_sigmoid_bias=0.0).training_logsnr_tables(t)
    assert not torch.allclose(base[0], other[0]) and not torch.allclose(base[3], other[3])
    torch.testing.assert_close(base[1] ** 2 + base[2] ** 2, torch.ones_like(base[1]))
    with pytest.raises(ValueError):
        DiffusionConfig(training_schedule_name="cosine_interpolated").training_logsnr_tables(t)
    with pytest.raises(ValueError):
        DiffusionConfig(loss_weighting_strategy="min_snr").training_logsnr_tables(t)
    assert lr_at_step(0, 5e-5, "constant_with_warmup", 10) == 0.0       # the reference's first optimizer step runs at lr 0
    assert lr_at_step(1, 5e-5, "constant_with_warmup", 10) == pytest.approx(5e-6)
    assert lr_at_step(10, 5e-5, "constant_with_warmup", 10) == pytest.approx(5e-5)
    assert lr_at_step(500, 5e-5, "constant_with_warmup", 10) == pytest.approx(5e-5)
    # the scheduler the reference builds (transformers.get_scheduler, stepped once after every optimizer step)
    import transformers
    for name, total in (("constant_with_warmup", None), ("linear", 40), ("cosine", 40)):
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=5e-5)
        sched = transformers.get_scheduler(name=name, optimizer=opt, num_warmup_steps=10, num_training_steps=total)
        for s in range(40):
            assert opt.param_groups[0]["lr"] == pytest.approx(lr_at_step(s, 5e-5, name, 10, total or 0), rel=1e-6, abs=1e-12), (name, s)
            opt.step()
            sched.step()
    assert lr_at_step(55, 1.0, "linear", 10, 100) == pytest.approx(0.5)
    assert lr_at_step(55, 1.0, "cosine", 10, 100) == pytest.approx(0.5)
    # the reference's scheduler goes through accelerator.prepare (simple_video_generation.py:183) -> AcceleratedScheduler steps the
    # wrapped scheduler num_processes times per optimizer step.  accelerate's own class, with the process count it would read on 3 GPUs
    import types
    from accelerate import scheduler as asched
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=5e-5)
    opt.step_was_skipped = False  # what AcceleratedOptimizer exposes
    inner = transformers.get_scheduler(name="constant_with_warmup", optimizer=opt, num_warmup_steps=10)
    wrapped = asched.AcceleratedScheduler.__new__(asched.AcceleratedScheduler)
    wrapped.scheduler, wrapped.optimizers, wrapped.split_batches, wrapped.step_with_optimizer = inner, [opt], False, True
    wrapped.gradient_state = types.SimpleNamespace(sync_gradients=True, adjust_scheduler=True)
    real_state, asched.AcceleratorState = asched.AcceleratorState, (lambda: types.SimpleNamespace(num_processes=3))
    try:
        for s in range(8):
            assert opt.param_groups[0]["lr"] == pytest.approx(lr_at_step(s, 5e-5, "constant_with_warmup", 10, num_processes=3), rel=1e-6, abs=1e-12), s
            opt.step()
            wrapped.step()
    finally:
        asched.AcceleratorState = real_state
    assert lr_at_step(4, 5e-5, "constant_with_warmup", 10, num_processes=3) == pytest.approx(5e-5)  # warm-up over after ceil(10 / 3) steps
