"""Loader that executes the reference's OWN source files on CPU (build container only).

Used only by tools/make_golden.py to capture golden vectors; never shipped to or run on the
GPU box (``/root/reference`` does not exist there) and never imported by the product or tests.

The reference imports third-party packages this image lacks.  They are replaced by minimal
stand-ins that restate the pinned upstream semantics (versions from the reference's
requirements.txt).  Each stand-in is a few lines; none of the reference's own code is
replaced or copied:

  timm==1.0.17     PatchEmbed  -> Conv2d(k=s=patch) [+ flatten(2).transpose(1,2) if flatten]
                   Mlp         -> fc1 -> act -> fc2
  diffusers==0.32.2 TimestepEmbedding -> linear_1 -> SiLU -> linear_2 ; LabelEmbedding -> nn.Embedding
  rotary_embedding_torch==0.8.6 rotate_half -> interleaved pairs (x1,x2) -> (-x2,x1)
  omegaconf==2.3.0 DictConfig  -> attribute dict with .get ; OmegaConf.to_container -> plain dict
  lightning==2.5.1 LightningModule -> nn.Module with .device/.log/.trainer
  wandb, roma, colorama, torchmetrics...: names only (never called on this path)

The reference's heavy package __init__ files (they import VAE / metric / dataset stacks) are
bypassed by registering namespace packages for ``algorithms``, ``algorithms.dfot`` ... so
that only the files on the hot path are executed.
"""
from __future__ import annotations

import importlib
import importlib.machinery
import sys
import types

import torch
from torch import nn

REF = "/root/reference"


def _mod(name: str, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _ns(name: str, path: str):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


class AttrDict(dict):
    """Stand-in for omegaconf.DictConfig: recursive attribute access + .get."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in {**(d or {}), **kw}.items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return cls(v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = self._wrap(v)


def _to_container(cfg, resolve=True):
    if isinstance(cfg, dict):
        return {k: _to_container(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [_to_container(v) for v in cfg]
    return cfg


class _PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None,
                 flatten=True, bias=True, **_):
        super().__init__()
        ps = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.patch_size = ps
        if img_size is not None:
            im = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
            self.grid_size = (im[0] // ps[0], im[1] // ps[1])
            self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=ps, stride=ps, bias=bias)
        self.norm = nn.Identity()

    def forward(self, x):
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)
        return self.norm(x)


class _Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU,
                 bias=True, drop=0.0, **_):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, **_):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


class _LabelEmbedding(nn.Module):
    def __init__(self, num_classes, hidden_size, dropout_prob):
        super().__init__()
        self.embedding_table = nn.Embedding(num_classes + int(dropout_prob > 0), hidden_size)

    def forward(self, labels, force_drop_ids=None):
        return self.embedding_table(labels)


def _rotate_half(x):
    x = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


class _LightningModule(nn.Module):
    trainer = None
    logger = None

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    def log(self, *a, **k):
        pass

    def log_dict(self, *a, **k):
        pass

    def save_hyperparameters(self, *a, **k):
        pass


def install():
    """Registers stand-ins + namespace packages, imports the hot-path reference modules and
    returns a dict of the reference classes needed."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import accelerate  # noqa: F401  (real, must precede the wandb stub)
    import transformers  # noqa: F401

    oc = _mod("omegaconf", DictConfig=AttrDict,
              OmegaConf=types.SimpleNamespace(to_container=_to_container, create=AttrDict))
    _mod("omegaconf.omegaconf", open_dict=lambda cfg: __import__("contextlib").nullcontext())
    oc.omegaconf = sys.modules["omegaconf.omegaconf"]
    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.vision_transformer", PatchEmbed=_PatchEmbed, Mlp=_Mlp, Attention=nn.Module)
    _mod("timm.layers", PatchEmbed=_PatchEmbed, Mlp=_Mlp)
    _mod("diffusers")
    _mod("diffusers.models")
    _mod("diffusers.models.embeddings", TimestepEmbedding=_TimestepEmbedding, LabelEmbedding=_LabelEmbedding)
    _mod("rotary_embedding_torch")
    _mod("rotary_embedding_torch.rotary_embedding_torch", rotate_half=_rotate_half)
    _mod("roma")
    _mod("wandb", Video=object)
    _mod("colorama", Fore=types.SimpleNamespace(CYAN="", RESET="", RED="", GREEN="", YELLOW=""))
    lt = _mod("lightning")
    pl = _mod("lightning.pytorch", LightningModule=_LightningModule)
    lt.pytorch = pl
    _mod("lightning.pytorch.utilities", grad_norm=lambda *a, **k: {})
    _mod("lightning.pytorch.utilities.types", STEP_OUTPUT=object)
    _mod("lightning.pytorch.loggers")
    _mod("lightning.pytorch.loggers.logger", Logger=object)
    _mod("lightning_utilities")
    _mod("lightning_utilities.core")
    _mod("lightning_utilities.core.apply_func", apply_to_collection=lambda d, *a, **k: d)

    for name, rel in [
        ("algorithms", "algorithms"), ("algorithms.common", "algorithms/common"),
        ("algorithms.dfot", "algorithms/dfot"), ("algorithms.dfot.backbones", "algorithms/dfot/backbones"),
        ("algorithms.dfot.backbones.modules", "algorithms/dfot/backbones/modules"),
        ("algorithms.dfot.backbones.u_vit", "algorithms/dfot/backbones/u_vit"),
        ("algorithms.dfot.backbones.dit", "algorithms/dfot/backbones/dit"),
        ("algorithms.dfot.diffusion", "algorithms/dfot/diffusion"), ("utils", "utils"),
    ]:
        _ns(name, f"{REF}/{rel}")
    _mod("utils.distributed_utils", is_rank_zero=True, rank_zero_print=print)
    _mod("utils.logging_utils", log_video=lambda *a, **k: None)
    dummy = type("Dummy", (), {})
    _mod("algorithms.vae", ImageVAE=dummy, VideoVAE=dummy, MyAutoencoderDC=dummy, AutoencoderKL=dummy,
         TiTok_KL=dummy)
    _mod("algorithms.common.metrics")
    _mod("algorithms.common.metrics.video", VideoMetric=dummy, SharedVideoMetricModelRegistry=dummy)
    _mod("algorithms.common.attn_hook", register_hooks=None, clear_hooks=None, save_attention_maps=None,
         attn_maps={})

    imp = importlib.import_module
    imp("algorithms.dfot.backbones.modules.embeddings")
    imp("algorithms.dfot.backbones.base_backbone")
    blocks = imp("algorithms.dfot.backbones.u_vit.u_vit_blocks")
    uvit = imp("algorithms.dfot.backbones.u_vit.u_vit3d")
    uvit_pose = imp("algorithms.dfot.backbones.u_vit.u_vit3d_pose")
    bb = sys.modules["algorithms.dfot.backbones"]
    for n in ("Unet3D", "DiT3D", "DiT3DPose", "FARDiT", "DIT1D", "DifferenceDiT3D"):
        setattr(bb, n, None)
    bb.UViT3D, bb.UViT3DPose = uvit.UViT3D, uvit_pose.UViT3DPose
    dit3d = imp("algorithms.dfot.backbones.dit.dit3d")  # K600 backbone (only the "full" variant is exercised)
    bb.DiT3D = dit3d.DiT3D
    diffdit = imp("algorithms.dfot.backbones.dit.difference_dit3d")  # bash/k600 backbone (factorized matrix attention)
    bb.DifferenceDiT3D = diffdit.DifferenceDiT3D
    dd = imp("algorithms.dfot.diffusion.discrete_diffusion")
    cd = imp("algorithms.dfot.diffusion.continuous_diffusion")
    dpk = sys.modules["algorithms.dfot.diffusion"]
    dpk.DiscreteDiffusion, dpk.ContinuousDiffusion = dd.DiscreteDiffusion, cd.ContinuousDiffusion
    hgm = imp("algorithms.dfot.history_guidance")
    pose_algo = imp("algorithms.dfot.dfot_video_pose")
    video_algo = sys.modules["algorithms.dfot.dfot_video"]
    diff_algo = imp("algorithms.dfot.difference_dfot_video")
    geo = imp("utils.geometry_utils")
    return {
        "UViT3DPose": uvit_pose.UViT3DPose, "blocks": blocks, "DiscreteDiffusion": dd.DiscreteDiffusion,
        "ContinuousDiffusion": cd.ContinuousDiffusion, "HistoryGuidance": hgm.HistoryGuidance,
        "DFoTVideoPose": pose_algo.DFoTVideoPose, "DFoTVideo": video_algo.DFoTVideo, "DiT3D": dit3d.DiT3D,
        "DifferenceDiT3D": diffdit.DifferenceDiT3D, "DifferenceDFoTVideo": diff_algo.DifferenceDFoTVideo,
        "geometry": geo, "AttrDict": AttrDict,
    }


def install_vae():
    """The reference's VideoVAE (algorithms/vae/video_vae/model.py and algorithms/vae/common/modules/*) importable on its own: namespace
    packages skip algorithms/vae/__init__.py and video_vae/__init__.py (they pull trainers / lightning / lpips), utils.ckpt_utils (wandb /
    huggingface download helpers, only used by from_pretrained) is a four-name stand-in.  Returns the VideoVAE class."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    for name, rel in [("algorithms", "algorithms"), ("algorithms.vae", "algorithms/vae"), ("algorithms.vae.video_vae", "algorithms/vae/video_vae"),
                      ("utils", "utils")]:
        _ns(name, f"{REF}/{rel}")
    _mod("utils.ckpt_utils", is_wandb_run_path=lambda p: False, is_hf_path=lambda p: False, wandb_to_local_path=lambda p: p,
         download_pretrained=lambda p: p)
    model = importlib.import_module("algorithms.vae.video_vae.model")   # imports algorithms.vae.common(.modules) for real
    return model.VideoVAE
