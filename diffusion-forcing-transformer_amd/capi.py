"""ctypes binding of libdfot_hip.so (include/dfot_hip.h).

There is no CPU fallback: importing this module raises if the HIP library has not been
built (``python -c "import __graft_entry__ as g; g.build()"``), and every call raises
``DfotError`` with the library's message when a status code is non-zero.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DFOT_LIB", os.path.join(_HERE, "libdfot_hip.so"))  # DFOT_LIB: A/B a differently built library

OK, ERR_ARG, ERR_SHAPE, ERR_HIP, ERR_STATE, ERR_NAME = range(6)


class DfotError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libdfot_hip error {code}: {message}")
        self.code = code


class UViTConfig(C.Structure):
    _fields_ = [
        ("channels", C.c_int32 * 4), ("emb_channels", C.c_int32), ("num_updown_blocks", C.c_int32 * 3),
        ("num_mid_blocks", C.c_int32), ("num_heads", C.c_int32), ("in_channels", C.c_int32),
        ("resolution", C.c_int32), ("max_tokens", C.c_int32), ("cond_dim", C.c_int32), ("noise_dim", C.c_int32),
        ("rope_theta", C.c_float), ("eps", C.c_float),
    ]


class DiTConfig(C.Structure):
    _fields_ = [
        ("hidden_size", C.c_int32), ("depth", C.c_int32), ("num_heads", C.c_int32), ("patch_size", C.c_int32),
        ("in_channels", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("max_tokens", C.c_int32),
        ("mlp_hidden", C.c_int32), ("noise_dim", C.c_int32), ("timesteps", C.c_int32),
        ("rope_theta", C.c_float), ("eps", C.c_float),
        ("variant", C.c_int32), ("embed_col_dim", C.c_int32), ("num_col_heads", C.c_int32), ("num_row_heads", C.c_int32),
        ("temporal_mlp_hidden", C.c_int32), ("use_bias", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/dfot_hip.h declares
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
SIGNATURES = {
    "dfot_last_error": (C.c_char_p, []),
    "dfot_version": (_I, []),
    "dfot_uvit_create": (_I, [C.POINTER(UViTConfig), C.POINTER(_P)]),
    "dfot_uvit_destroy": (_I, [_P]),
    "dfot_uvit_num_params": (_I, [_P]),
    "dfot_uvit_param_name": (C.c_char_p, [_P, _I]),
    "dfot_uvit_param_shape": (_I, [_P, _I, C.POINTER(_L), C.POINTER(_I)]),
    "dfot_uvit_load_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(_L), _I, _P]),
    "dfot_uvit_finalize": (_I, [_P, _P]),
    "dfot_uvit_reserve": (_I, [_P, _I]),
    "dfot_uvit_workspace_bytes": (C.c_size_t, [_P]),
    "dfot_uvit_set_option": (_I, [_P, C.c_char_p, _I]),
    "dfot_uvit_query": (_I, [_P, C.c_char_p, C.POINTER(C.c_double)]),
    "dfot_uvit_attn_timing": (_I, [_P, C.POINTER(C.c_double), C.POINTER(_L)]),
    "dfot_uvit_forward": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "dfot_uvit_set_conditions": (_I, [_P, _P, _P, _I, _P]),
    "dfot_uvit_forward_cached": (_I, [_P, _P, _P, _P, _I, _P]),
    "dfot_uvit_forward_cached_live": (_I, [_P, _P, _P, _P, _I, _P, _P]),
    "dfot_uvit_forward_cached_masks": (_I, [_P, _P, _P, _P, _I, _P, _P, _P]),
    "dfot_uvit_read_tap": (_I, [_P, C.c_char_p, _P, C.c_size_t, _P]),
    "dfot_dit_create": (_I, [C.POINTER(DiTConfig), C.POINTER(_P)]),
    "dfot_dit_destroy": (_I, [_P]),
    "dfot_dit_num_params": (_I, [_P]),
    "dfot_dit_param_name": (C.c_char_p, [_P, _I]),
    "dfot_dit_param_shape": (_I, [_P, _I, C.POINTER(_L), C.POINTER(_I)]),
    "dfot_dit_load_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(_L), _I, _P]),
    "dfot_dit_finalize": (_I, [_P, _P]),
    "dfot_dit_reserve": (_I, [_P, _I]),
    "dfot_dit_workspace_bytes": (C.c_size_t, [_P]),
    "dfot_dit_set_option": (_I, [_P, C.c_char_p, _I]),
    "dfot_dit_attn_timing": (_I, [_P, C.POINTER(C.c_double), C.POINTER(_L)]),
    "dfot_dit_forward": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "dfot_dit_read_tap": (_I, [_P, C.c_char_p, _P, C.c_size_t, _P]),
    "dfot_dit_train_create": (_I, [C.POINTER(DiTConfig), C.POINTER(_P)]),
    "dfot_dit_train_destroy": (_I, [_P]),
    "dfot_dit_train_num_params": (_I, [_P]),
    "dfot_dit_train_param_name": (C.c_char_p, [_P, _I]),
    "dfot_dit_train_param_shape": (_I, [_P, _I, C.POINTER(_L), C.POINTER(_I)]),
    "dfot_dit_train_param_offset": (_L, [_P, _I]),
    "dfot_dit_train_total_numel": (_L, [_P]),
    "dfot_dit_train_workspace_bytes": (C.c_size_t, [_P]),
    "dfot_dit_train_attach": (_I, [_P, _P, _P]),
    "dfot_dit_train_reserve": (_I, [_P, _I]),
    "dfot_dit_train_sync_weights": (_I, [_P, _P]),
    "dfot_dit_train_forward": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "dfot_dit_train_input_grad": (_I, [_P, _P, _P]),
    "dfot_dit_train_backward": (_I, [_P, _P, _P]),
    "dfot_vloss_grad": (_I, [_P] * 7 + [_I, _I, _L, _I, _P]),
    "dfot_sumsq": (_I, [_P, _L, _P, _P]),
    "dfot_adamw_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P, _F, _P, _F, _P]),
    "dfot_ray_encode": (_I, [_P, _P, _I, _I, _I, _P]),
    "dfot_ray_encode_normalized": (_I, [_P, _P, _I, _I, _I, _P]),
    "dfot_hg_prepare": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _L, _P]),
    "dfot_ddim_compose": (_I, [_P] * 11 + [_I, _I, _I, _L, _P]),
    "dfot_ddim_compose_tokw": (_I, [_P] * 11 + [_I, _I, _I, _L, _P]),
    "dfot_ddim_noise": (_I, [_P] * 5 + [_I, _I, _I, _L, _I, _P]),
    "dfot_vpred_loss": (_I, [_P] * 9 + [_I, _I, _L, _P]),
    "dfot_vpred_loss_scratch_floats": (_L, [_I, _I, _L]),
    "dfot_vspace_loss": (_I, [_P] * 9 + [_I, _I, _L, _P]),
    "dfot_op_gemm": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_conv3x3": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfot_op_attention": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfot_op_attention_padded": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfot_op_attention_bwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_conv3x3_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfot_op_conv3x3_bwd2": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfot_op_gn_silu_bwd": (_I, [_P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dfot_op_rms_film_bwd": (_I, [_P, _P, _P, _P, _F, _P, _P, _P, _L, _I, _I, _P]),
    "dfot_op_rms_film_bwd_res": (_I, [_P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _L, _I, _P]),
    "dfot_op_qknorm_rope_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _F, _P, _I, _P, _P, _L, _I, _I, _I, _P]),
    "dfot_op_wgrad_nt": (_I, [_P, _I, _P, _I, _P, _I, _I, _L, _I, _P]),
    "dfot_op_gemm_bf16": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_gemm_f32": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_transpose_bf16": (_I, [_P, _P, _I, _I, _P]),
    "dfot_op_colsum_bf16": (_I, [_P, _I, _P, _L, _I, _P]),
    "dfot_op_rms_film_fwd": (_I, [_P, _P, _P, _F, _P, _L, _I, _P]),
    "dfot_op_qknorm_rope_fwd": (_I, [_P, _I, _P, _P, _P, _F, _F, _P, _P, _P, _L, _I, _I, _I, _P]),
    "dfot_op_fused_proj_train": (_I, [_P, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _I, _I, _L, _I, _I, _I, _P]),
    "dfot_op_silu_cols": (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _L, _I, _P]),
    "dfot_op_attention_fwd_lse": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "dfot_op_attention_fwd_lse_bounded": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, C.c_float, _P, C.c_size_t, _P]),
    "dfot_op_attention_scratch_bytes": (C.c_size_t, [_I, _I, _I, _I]),
    "dfot_op_attention_bwd_lse": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_gn_silu_fwd": (_I, [_P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _P]),
    "dfot_op_gn_silu_bwd5": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _P]),
    "dfot_op_gn_silu_fwd2": (_I, [_P, _P, _P, _P, _L, _F, _P, _P, _I, _I, _I, _P]),
    "dfot_op_gn_silu_bwd6": (_I, [_P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _P]),
    "dfot_op_gemm_bf16_frame_bias": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "dfot_op_split_bf16": (_I, [_P, _P, _P, _L, _P]),
    "dfot_op_frame_sums_bf16": (_I, [_P, _L, _P, _I, _I, _I, _P]),
    "dfot_op_sgemm": (_I, [_P, _L, _L, _P, _L, _L, _P, _L, _I, _I, _I, _I, _P]),
    "dfot_op_pack_conv3": (_I, [_P, _P, _I, _I, _I, _P]),
    "dfot_op_conv3x3_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfot_op_pool2_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_pool2_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_sub_bf16": (_I, [_P, _P, _P, _L, _P]),
    "dfot_op_upsample_add": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_upsample_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_axpy": (_I, [_P, _P, _F, _L, _P]),
    "dfot_op_mul_cols": (_I, [_P, _I, _I, _P, _L, _I, _P]),
    "dfot_op_emb_pyramid": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "dfot_op_cond_repack": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_embed_input": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_embed_input_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfot_op_embed_input_dgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dfot_op_project_output": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_outgrad_gather": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_groupnorm_scratch_floats": (_L, [_I, _I]),
    "dfot_op_groupnorm": (_I, [_P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _P]),
    "dfot_op_frame_shift": (_I, [_P, _P, _I, _I, _L, _I, _P]),
    "dfot_op_upsample3d": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dfot_op_softmax_rows": (_I, [_P, _P, _L, _I, _F, _P]),
    "dfot_op_f32_to_bf16": (_I, [_P, _P, _L, _P]),
    "dfot_op_bf16_to_f32": (_I, [_P, _P, _L, _P]),
}


def _load() -> C.CDLL:
    # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE libdfot_hip.so is
    # loaded, so that the library's libamdhip64 dependency resolves to the SAME runtime torch uses (one HIP runtime per
    # process: with two, the second one finds no device and device pointers / streams could not be shared anyway).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. There is no CPU fallback; "
            "run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


def check(code: int) -> None:
    if code != OK:
        raise DfotError(code, (lib.dfot_last_error() or b"").decode())


def ptr(t, dtype=None, name: str = "tensor") -> C.c_void_p:
    """Device pointer of a torch tensor (None -> NULL).  The kernels dereference it on the GPU, so a host tensor, a strided
    view or (when the caller states one) a wrong element type is refused HERE: a bad pointer would be a GPU memory fault."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise ValueError(f"{name} is on {t.device}: libdfot_hip takes GPU memory only (there is no CPU path)")
    if not t.is_contiguous():
        raise ValueError(f"{name} with shape {tuple(t.shape)} and strides {tuple(t.stride())} is not contiguous")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name} has dtype {t.dtype}, expected {dtype}")
    return C.c_void_p(t.data_ptr())


def ptr_rows(t, dtype=None, name: str = "tensor") -> C.c_void_p:
    """Device pointer of a 2-D matrix whose rows are contiguous but may be a column block of a wider one (stride(0) >= shape[1]): for
    the entry points that take a row stride next to the pointer.  Everything else as ptr()."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise ValueError(f"{name} is on {t.device}: libdfot_hip takes GPU memory only (there is no CPU path)")
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        raise ValueError(f"{name} with shape {tuple(t.shape)} and strides {tuple(t.stride())} is not a row-strided matrix")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name} has dtype {t.dtype}, expected {dtype}")
    return C.c_void_p(t.data_ptr())


def require_device(dev, **tensors) -> None:
    """Every given tensor (None entries skipped) must live on `dev`; raises ValueError naming the first that does not."""
    for name, t in tensors.items():
        if t is not None and t.device != dev:
            raise ValueError(f"{name} is on {t.device} but the backbone's parameters are on {dev}")


def stream_ptr() -> C.c_void_p:
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
