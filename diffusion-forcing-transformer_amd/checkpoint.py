"""Checkpoint ingestion for the HIP backbone.

Reads what the reference writes and maps it onto ``UViT3DPose.load_state_dict``:
  * Lightning ``.ckpt`` files: ``checkpoint["state_dict"]`` with keys ``diffusion_model.model.*`` (only those are
    kept, algorithms/common/base_pytorch_video_algo.py:1112-1125), optionally with the ``torch.compile`` prefix
    ``diffusion_model._orig_mod.model.*`` (:1096-1110);
  * EMA weights: ``optimizer_states[0]["ema"]`` is a list in ``diffusion_model.named_parameters()`` order that
    replaces the trained weights for inference unless the file is a released ``pretrained_ema`` checkpoint
    (:1185-1201); the accelerate loop stores them as ``ema.safetensors`` instead
    (experiments/simple_video_generation.py:613-619,653-657);
  * strict key check with the reference's error text (:1162-1182).
Files are opened with loaders that execute nothing from the file (``torch.load(weights_only=True)``, safetensors).
"""
from __future__ import annotations

import os
from typing import Dict, List, Mapping, Optional, Tuple

import torch

PREFIXES = ("diffusion_model._orig_mod.model.", "diffusion_model.model.")


def reference_parameter_order(model) -> List[str]:
    """Names in the order of the reference's ``named_parameters()``: UViT3D creates down_blocks and up_blocks before
    mid_blocks (u_vit3d.py:113-185), so up_blocks.* precede mid_blocks.*"""
    names = [k for k, _ in model.named_parameters()]
    head = [k for k in names if not k.startswith(("mid_blocks", "up_blocks"))]
    return head + [k for k in names if k.startswith("up_blocks")] + [k for k in names if k.startswith("mid_blocks")]


def read_checkpoint(path: str) -> Mapping:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return {"state_dict": load_file(path)}
    return torch.load(path, map_location="cpu", weights_only=True)


def extract_backbone_state(checkpoint: Mapping, model, use_ema: bool = True) -> Tuple[Dict[str, torch.Tensor], List[str]]:
    """Returns (backbone state dict with bare key names, ignored checkpoint keys)."""
    sd = checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint
    out: Dict[str, torch.Tensor] = {}
    ignored: List[str] = []
    own = set(model.state_dict().keys())
    for key, value in sd.items():
        prefix = next((p for p in PREFIXES if key.startswith(p)), None)
        if prefix is not None:
            out[key[len(prefix):]] = value
        elif key in own:  # bare backbone keys: ema.safetensors / a state dict saved from this module
            out[key] = value
        else:
            ignored.append(key)
    if use_ema and not checkpoint.get("pretrained_ema", False) and checkpoint.get("optimizer_states"):
        ema = checkpoint["optimizer_states"][0].get("ema")
        if ema is not None:
            # the reference zips diffusion_model.named_parameters() = ["model." + name ...] with the EMA list
            order = reference_parameter_order(model)
            assert len(order) == len(ema), "Number of original weights and EMA weights do not match."
            for name, w in zip(order, ema):
                out[name] = w
    return out, ignored


def load_reference_checkpoint(model, path_or_checkpoint, strict: bool = True, use_ema: bool = True) -> List[str]:
    ckpt = read_checkpoint(path_or_checkpoint) if isinstance(path_or_checkpoint, (str, os.PathLike)) else path_or_checkpoint
    state, ignored = extract_backbone_state(ckpt, model, use_ema=use_ema)
    expected = list(model.state_dict().keys())
    missing = [k for k in expected if k not in state]
    if missing and strict:
        raise ValueError(
            f"The following keys are not found in the checkpoint: {missing[:8]}{'...' if len(missing) > 8 else ''}. "
            "Thus, the checkpoint cannot be loaded. To ignore this error, turn off strict checkpoint loading.")
    current = model.state_dict()
    merged = {k: state.get(k, current[k]) for k in expected}
    model.load_state_dict(merged, strict=True)
    return ignored
