"""torch.library registration of the engine's entry points (namespace ``dfot``).

The reference is a PyTorch program; its natural plug-in surface is a set of torch operators.  Each operator below is a thin
shim over one C-ABI call of libdfot_hip.so (``capi``) with a fake ("meta") implementation for shape inference, so that
FakeTensor tracing, ``torch.compile`` graphs and stream capture treat the HIP kernels as opaque ops:

    dfot::uvit3d_pose_forward(x, noise_levels, external_cond, external_cond_mask?, model) -> v      [dfot_uvit_forward]
    dfot::uvit3d_pose_forward_train(x, noise_levels, external_cond, mask?, params[], model) -> v    [autograd: dfot_op_* forward/backward]
    dfot::dit3d_forward(x, noise_levels, model) -> v                                                [dfot_dit_forward]
    dfot::ray_encoding(raw_poses, resolution) -> cond                                               [dfot_ray_encode]
    dfot::hg_prepare(x, noise?, qa, qb, nfe) -> x_in                                                [dfot_hg_prepare]
    dfot::ddim_hg_step(x, x_in, v, sa, s1, an, cn, keep, weight, gen, nfe) -> x_next                [dfot_ddim_compose / _tokw]

``model`` is an integer key of a live backbone module (``register_model``): operators take tensors and scalars only.
The nn.Module mirrors (`UViT3DPose`, `DiT3D`, `DifferenceDiT3D`) dispatch their ``forward`` through these operators and the
sampler's ``_process_conditions`` through ``dfot::ray_encoding``; the sampler's step loop calls the same C entry points
directly (it re-uses its output buffers across steps), ``dfot::hg_prepare`` / ``dfot::ddim_hg_step`` are the functional forms
for a host that drives the step itself.
"""
from __future__ import annotations

import weakref
from typing import List, Optional

import torch
from torch import Tensor
from torch.library import custom_op

from . import capi

_MODELS: "weakref.WeakValueDictionary[int, torch.nn.Module]" = weakref.WeakValueDictionary()


def register_model(module: torch.nn.Module) -> int:
    key = id(module)
    _MODELS[key] = module
    return key


def _model(key: int):
    try:
        return _MODELS[key]
    except KeyError:
        raise RuntimeError(f"dfot: backbone {key} is not registered (or was garbage-collected)") from None


@custom_op("dfot::uvit3d_pose_forward", mutates_args=())
def uvit3d_pose_forward(x: Tensor, noise_levels: Tensor, external_cond: Tensor, external_cond_mask: Optional[Tensor],
                        model: int) -> Tensor:
    return _model(model)._forward_impl(x, noise_levels, external_cond, external_cond_mask)


@uvit3d_pose_forward.register_fake
def _(x, noise_levels, external_cond, external_cond_mask, model):
    return torch.empty_like(x)


# ---- training form: the same backbone under autograd -------------------------------------------------------------------
# `params` are the module's trainable tensors in `model._train_names` order; they are operator inputs so that autograd hands
# their gradients back (torch.library.register_autograd).  Forward = the saved-activation forward of uvit_train.UViT3DPoseTrainer
# on the module's current weights, backward = its hand-written backward: what `accelerator.backward(loss)` walks into when the
# reference's training_step calls `self.model(x_t, precond_scale * logsnr, external_cond)` (continuous_diffusion.py:154,
# experiments/simple_video_generation.py:260-270).  The gradient w.r.t. x is produced only when x requires it: training never
# differentiates w.r.t. x_t, reconstruction guidance does (discrete_diffusion.py:485-513: torch.autograd.grad(guidance_loss, x)).
@custom_op("dfot::uvit3d_pose_forward_train", mutates_args=())
def uvit3d_pose_forward_train(x: Tensor, noise_levels: Tensor, external_cond: Tensor, external_cond_mask: Optional[Tensor],
                              params: List[Tensor], model: int) -> Tensor:
    return _model(model)._train_forward_impl(x, noise_levels, external_cond, external_cond_mask, params)


@uvit3d_pose_forward_train.register_fake
def _(x, noise_levels, external_cond, external_cond_mask, params, model):
    return torch.empty_like(x)


# `stamp` = the module's training-forward counter at the time of the forward this backward belongs to: the saved activations live
# in the module's ONE engine, so a second training forward before this backward has overwritten them -- the backward then raises
# instead of returning the gradients of another input (forward, forward, backward, backward is refused; run backward after each
# forward, e.g. gradient accumulation as forward/backward pairs).
@custom_op("dfot::uvit3d_pose_backward", mutates_args=())
def uvit3d_pose_backward(grad_out: Tensor, params: List[Tensor], model: int, stamp: int, want_dx: bool = False) -> List[Tensor]:
    return _model(model)._train_backward_impl(grad_out, params, stamp, want_dx)


@uvit3d_pose_backward.register_fake
def _(grad_out, params, model, stamp, want_dx=False):
    return [torch.empty_like(p) for p in params] + ([torch.empty_like(grad_out)] if want_dx else [])


def _train_setup_context(ctx, inputs, output):
    ctx.params = inputs[4]
    ctx.model = inputs[5]
    ctx.stamp = _model(inputs[5])._train_stamp  # setup_context runs right after the forward it describes


def _train_backward(ctx, grad_out):
    want_dx = bool(ctx.needs_input_grad[0])
    grads = torch.ops.dfot.uvit3d_pose_backward(grad_out.contiguous(), ctx.params, ctx.model, ctx.stamp, want_dx)
    dx = None
    if want_dx:
        dx = grads[-1].view_as(grad_out)
        grads = grads[:-1]
    return dx, None, None, None, grads, None


uvit3d_pose_forward_train.register_autograd(_train_backward, setup_context=_train_setup_context)


@custom_op("dfot::dit3d_forward", mutates_args=())
def dit3d_forward(x: Tensor, noise_levels: Tensor, model: int) -> Tensor:
    return _model(model)._forward_impl(x, noise_levels)


@dit3d_forward.register_fake
def _(x, noise_levels, model):
    return torch.empty_like(x)


# DiT3D / DifferenceDiT3D under autograd: as uvit3d_pose_forward_train, backed by trainer.DiT3DTrainer (dfot_dit_train_*)
@custom_op("dfot::dit3d_forward_train", mutates_args=())
def dit3d_forward_train(x: Tensor, noise_levels: Tensor, params: List[Tensor], model: int) -> Tensor:
    return _model(model)._train_forward_impl(x, noise_levels, params)


@dit3d_forward_train.register_fake
def _(x, noise_levels, params, model):
    return torch.empty_like(x)


@custom_op("dfot::dit3d_backward", mutates_args=())
def dit3d_backward(grad_out: Tensor, params: List[Tensor], model: int, stamp: int, want_dx: bool = False) -> List[Tensor]:
    return _model(model)._train_backward_impl(grad_out, params, stamp, want_dx)


@dit3d_backward.register_fake
def _(grad_out, params, model, stamp, want_dx=False):
    return [torch.empty_like(p) for p in params] + ([torch.empty_like(grad_out)] if want_dx else [])


def _dit_train_setup_context(ctx, inputs, output):
    ctx.params = inputs[2]
    ctx.model = inputs[3]
    ctx.stamp = _model(inputs[3])._train_stamp


def _dit_train_backward(ctx, grad_out):
    want_dx = bool(ctx.needs_input_grad[0])
    grads = torch.ops.dfot.dit3d_backward(grad_out.contiguous(), ctx.params, ctx.model, ctx.stamp, want_dx)
    dx = None
    if want_dx:
        dx, grads = grads[-1], grads[:-1]
    return dx, None, grads, None


dit3d_forward_train.register_autograd(_dit_train_backward, setup_context=_dit_train_setup_context)


@custom_op("dfot::ray_encoding", mutates_args=())
def ray_encoding(raw_poses: Tensor, resolution: int, normalized: bool = False) -> Tensor:
    """normalized = False: poses are raw and become relative to frame 0 (the reference's default, normalize_by "first");
    True: the caller already expressed them in its world frame (pose.normalize_poses)"""
    b, t = raw_poses.shape[:2]
    raw = raw_poses.detach().to(device="cuda", dtype=torch.float32).contiguous()
    out = torch.empty(b, t, 180, resolution, resolution, device="cuda", dtype=torch.float32)
    fn = capi.lib.dfot_ray_encode_normalized if normalized else capi.lib.dfot_ray_encode
    capi.check(fn(capi.ptr(raw), capi.ptr(out), b, t, resolution, capi.stream_ptr()))
    return out


@ray_encoding.register_fake
def _(raw_poses, resolution, normalized=False):
    b, t = raw_poses.shape[:2]
    return raw_poses.new_empty((b, t, 180, resolution, resolution), dtype=torch.float32)


@custom_op("dfot::hg_prepare", mutates_args=())
def hg_prepare(x: Tensor, noise: Optional[Tensor], qa: Tensor, qb: Tensor, nfe: int) -> Tensor:
    b, t = x.shape[:2]
    f = x[0, 0].numel()
    F32 = torch.float32
    for name, tab in (("qa", qa), ("qb", qb)):
        if tuple(tab.shape) != (b * nfe, t):
            raise ValueError(f"{name} has shape {tuple(tab.shape)}, expected {(b * nfe, t)}")
    if noise is not None and tuple(noise.shape) != (b * nfe, *x.shape[1:]):
        raise ValueError(f"noise has shape {tuple(noise.shape)}, expected {(b * nfe, *x.shape[1:])}")
    px, pn, pa, pb = capi.ptr(x, F32, "x"), capi.ptr(noise, F32, "noise"), capi.ptr(qa, F32, "qa"), capi.ptr(qb, F32, "qb")
    x_in = torch.empty(b * nfe, *x.shape[1:], device=x.device, dtype=torch.float32)
    capi.check(capi.lib.dfot_hg_prepare(px, pn, pa, pb, capi.ptr(x_in), b, nfe, t, f, capi.stream_ptr()))
    return x_in


@hg_prepare.register_fake
def _(x, noise, qa, qb, nfe):
    return x.new_empty((x.shape[0] * nfe, *x.shape[1:]))


@custom_op("dfot::ddim_hg_step", mutates_args=())
def ddim_hg_step(x: Tensor, x_in: Tensor, v: Tensor, sa: Tensor, s1: Tensor, an: Tensor, cn: Tensor, keep: Tensor,
                 weight: Tensor, gen: Tensor, nfe: int) -> Tensor:
    b, t = x.shape[:2]
    f = x[0, 0].numel()
    F32 = torch.float32
    for name, big in (("x_in", x_in), ("v", v)):
        if tuple(big.shape) != (b * nfe, *x.shape[1:]):
            raise ValueError(f"{name} has shape {tuple(big.shape)}, expected {(b * nfe, *x.shape[1:])}")
    tabs = dict(sa=sa, s1=s1, an=an, cn=cn, keep=keep)
    for name, tab in tabs.items():
        if tuple(tab.shape) != (b * nfe, t):
            raise ValueError(f"{name} has shape {tuple(tab.shape)}, expected {(b * nfe, t)}")
    if tuple(gen.shape) != (b, t):
        raise ValueError(f"gen has shape {tuple(gen.shape)}, expected {(b, t)}")
    tokw = weight.ndim == 2
    if (tokw and tuple(weight.shape) != (nfe, t)) or (not tokw and tuple(weight.shape) != (nfe,)):
        raise ValueError(f"weight has shape {tuple(weight.shape)}, expected {(nfe,)} or {(nfe, t)}")
    ptrs = [capi.ptr(x, F32, "x"), capi.ptr(x_in, F32, "x_in"), capi.ptr(v, F32, "v")]
    ptrs += [capi.ptr(tab, F32, name) for name, tab in tabs.items()]
    ptrs += [capi.ptr(weight, F32, "weight"), capi.ptr(gen, torch.uint8, "gen")]
    x_next = torch.empty_like(x)
    fn = capi.lib.dfot_ddim_compose_tokw if tokw else capi.lib.dfot_ddim_compose
    capi.check(fn(*ptrs, capi.ptr(x_next), b, nfe, t, f, capi.stream_ptr()))
    return x_next


@ddim_hg_step.register_fake
def _(x, x_in, v, sa, s1, an, cn, keep, weight, gen, nfe):
    return torch.empty_like(x)
