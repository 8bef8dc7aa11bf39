"""ctypes binding of the C-ABI library (include/savit.h).  There is NO fallback: if libsavit.so is
missing or a symbol is absent, importing callers get a RuntimeError - the product path never routes
through PyTorch eager math or the CPU oracle."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsavit.so")

SAVIT_EINVAL = 1001
ABI_VERSION = 2  # include/savit.h SAVIT_ABI_VERSION: the struct layouts below belong to this version

EPI_BF16, EPI_BIAS_GELU, EPI_RESID, EPI_DGELU, EPI_F32, EPI_PATCH = range(6)


class GemmArgs(Structure):
    """Mirror of `savit_gemm_args` (include/savit.h)."""
    _fields_ = [
        ("A", c_void_p), ("Bt", c_void_p), ("C", c_void_p), ("C2", c_void_p), ("bias", c_void_p), ("aux", c_void_p),
        ("colscale", c_void_p), ("rowscale", c_void_p), ("colsum", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_int), ("ldb", c_int), ("ldc", c_int), ("ldaux", c_int),
        ("epilogue", c_int), ("alpha", c_float), ("alpha_cols", c_int), ("rows_per_sample", c_int),
        ("round_out_bf16", c_int), ("round_bias_bf16", c_int),
        ("img_size", c_int), ("patch", c_int), ("tokens", c_int), ("token_offset", c_int),
        ("tile", c_int), ("colsum_rows", c_int), ("cu_budget", c_int),
    ]


class WgradProblem(Structure):
    """Mirror of `savit_wgrad_problem` (include/savit.h)."""
    _fields_ = [("X", c_void_p), ("dY", c_void_p), ("dW", c_void_p), ("M", c_int), ("Kin", c_int), ("Nout", c_int), ("ldx", c_int),
                ("lddy", c_int), ("lddw", c_int), ("tile_begin", c_int), ("tile_count", c_int), ("overwrite", c_int)]


class ColsumJob(Structure):
    """Mirror of `savit_colsum_job` (include/savit.h)."""
    _fields_ = [("partial", c_void_p), ("nblk", c_int), ("d", c_int), ("nf", c_int), ("out", c_void_p * 4), ("extra_slab", c_void_p),
                ("extra_rows", c_int), ("extra_n", c_int), ("extra_out", c_void_p)]


class TransposeJob(Structure):
    """Mirror of `savit_transpose_job` (include/savit.h)."""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("src_batch_stride", c_long), ("dst_batch_stride", c_long), ("ld_src", c_int),
                ("ld_dst", c_int), ("batch", c_int), ("rows", c_int), ("cols", c_int)]


class GemmF32Args(Structure):
    """Mirror of `savit_gemm_f32_args` (include/savit.h)."""
    _fields_ = [("A", c_void_p), ("W", c_void_p), ("C", c_void_p), ("bias", c_void_p), ("aux", c_void_p), ("colscale", c_void_p),
                ("rowscale", c_void_p), ("C2", c_void_p), ("U", c_void_p),
                ("M", c_int), ("N", c_int), ("K", c_int), ("lda", c_int), ("ldw", c_int), ("ldc", c_int), ("ldaux", c_int),
                ("transA", c_int), ("transW", c_int), ("batch", c_int), ("inner", c_int),
                ("sAo", c_long), ("sAi", c_long), ("sWo", c_long), ("sWi", c_long), ("sCo", c_long), ("sCi", c_long),
                ("alpha", c_float), ("alpha_cols", c_int), ("act", c_int), ("accumulate", c_int), ("rows_per_sample", c_int),
                ("aux_row_mod", c_int), ("rowbias", c_void_p)]


# name -> (restype, argtypes); every symbol include/savit.h declares must be here (tests check both ways)
_SIGNATURES = {
    "savit_abi_version": (c_int, []),
    "savit_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_long, c_float,
                                    c_int, c_void_p]),
    "savit_layernorm_bwd": (c_int, [c_void_p] * 11 + [c_int, c_int, c_long, c_long, c_int, c_void_p, c_long, c_void_p]),
    "savit_layernorm_bwd_ex": (c_int, [c_void_p] * 11 + [c_int, c_int, c_long, c_long, c_int, c_void_p, c_long, c_void_p, c_int, c_int, c_void_p,
                                       c_void_p]),
    "savit_layernorm_bwd_sparse": (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_int, c_long] + [c_void_p] * 5 + [c_int, c_int, c_long, c_long, c_int, c_void_p,
                                           c_long, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "savit_layernorm_bwd_workspace_bytes": (c_long, [c_int, c_int]),
    "savit_layernorm_bwd_grid": (c_int, [c_int]),
    "savit_layernorm_bwd_finalize_jobs": (c_int, [POINTER(ColsumJob), c_int, c_void_p]),
    "savit_layernorm_fwd_mapped": (c_int, [c_void_p] * 6 + [c_int, c_int, c_long, c_float, c_int, c_int, c_int, c_int, c_void_p]),
    "savit_layernorm_bwd_mapped": (c_int, [c_void_p] * 11 + [c_int, c_int, c_long, c_long, c_int, c_int, c_int, c_int, c_void_p, c_long, c_void_p]),
    "savit_layerscale_bwd": (c_int, [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_long, c_void_p, c_long, c_void_p]),
    "savit_layernorm_bwd_ls": (c_int, [c_void_p] * 9 + [c_int, c_int, c_long, c_long, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_long, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "savit_class_attention_fwd": (c_int, [c_void_p, c_long, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "savit_cls_query_attention_fwd": (c_int, [c_void_p, c_long, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "savit_cls_query_attention_bwd": (c_int, [c_void_p, c_long, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_int, c_int,
                                              c_int, c_int, c_float, c_void_p]),
    "savit_class_attention_bwd": (c_int, [c_void_p, c_long, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_int, c_int, c_int,
                                          c_int, c_float, c_void_p]),
    "savit_gemm_bf16_tn": (c_int, [POINTER(GemmArgs), c_void_p]),
    "savit_gemm_tn_auto_tile": (c_int, [c_int, c_int, c_int]),
    "savit_gemm_tn_auto_tile_epi": (c_int, [c_int, c_int, c_int, c_int]),
    "savit_gemm_tn_auto_tile_cus": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "savit_gemm_colsum_rows": (c_int, [c_int, c_int, c_int, c_int]),
    "savit_gemm_colsum_rows_cus": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "savit_colsum_finalize": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "savit_gemm_wgrad_auto_variant": (c_int, [c_int, c_int, c_int]),
    "savit_gemm_bf16_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_void_p]),
    "savit_gemm_bf16_wgrad_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_int, c_void_p, c_long, c_void_p]),
    "savit_gemm_wgrad_workspace_bytes": (c_long, [c_int, c_int, c_int, c_int, c_int]),
    "savit_gemm_bf16_wgrad_partial": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                              c_void_p, c_long, c_void_p]),
    "savit_gemm_wgrad_reduce": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "savit_gemm_wgrad_split_count": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "savit_gemm_bf16_wgrad_grouped": (c_int, [POINTER(WgradProblem), c_int, c_int, c_void_p]),
    "savit_gemm_bf16_wgrad_grouped_ex": (c_int, [POINTER(WgradProblem), c_int, c_int, c_void_p, c_void_p]),
    "savit_gemm_wgrad_group_tiles": (c_int, [c_int, c_int, c_int]),
    "savit_attention_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "savit_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                    c_void_p]),
    "savit_th_attention_fwd": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p]),
    "savit_th_attention_bwd": (c_int, [c_void_p] * 10 + [c_int] * 6 + [c_float, c_void_p, c_long, c_void_p]),
    "savit_th_attention_bwd_workspace_bytes": (c_long, [c_int, c_int, c_int]),
    "savit_th_fused_supported": (c_int, [c_int, c_int, c_int]),
    "savit_th_fused_preferred": (c_int, [c_int, c_int, c_int]),
    "savit_th_fused_fwd_workspace_bytes": (c_long, [c_int, c_int, c_int, c_int]),
    "savit_th_fused_bwd_workspace_bytes": (c_long, [c_int, c_int, c_int, c_int]),
    "savit_th_fused_attention_fwd": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p, c_long, c_void_p]),
    "savit_th_fused_attention_bwd": (c_int, [c_void_p] * 9 + [c_int] * 6 + [c_float, c_void_p, c_long, c_void_p]),
    "savit_th_attention_bwd_products": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_float, c_void_p]),
    "savit_cls_pos_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_void_p]),
    "savit_pos_cls_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "savit_transpose_bf16": (c_int, [c_void_p, c_long, c_int, c_void_p, c_long, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                     c_void_p, c_int, c_void_p]),
    "savit_transpose_bf16_jobs": (c_int, [POINTER(TransposeJob), c_int, c_void_p]),
    "savit_transpose_rowsum_rows": (c_int, [c_int, c_int]),
    "savit_token_mean_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "savit_token_mean_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "savit_seq16_attention_fwd": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "savit_seq16_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "savit_tnt_pixel_gather": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "savit_add_rows_periodic": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_void_p]),
    "savit_tnt_inner2outer_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "savit_tnt_inner2outer_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "savit_cast_colsum": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "savit_gather_rows_bf16": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_void_p]),
    "savit_scatter_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p]),
    "savit_softmax_xent": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                   c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "savit_sumsq": (c_int, [c_void_p, c_long, c_void_p, c_void_p]),
    "savit_sumsq_ranges": (c_int, [c_void_p, POINTER(c_long), c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "savit_zero_ranges": (c_int, [c_void_p, POINTER(c_long), c_int, c_void_p]),
    "savit_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float, c_float, c_float, c_int,
                                 c_void_p, c_float, c_float, c_void_p]),
    "savit_adamw_step_mirror": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float, c_float, c_float, c_int,
                                        c_void_p, c_float, c_float, c_void_p, c_void_p]),
    "savit_cast_transpose_bf16": (c_int, [c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_long, c_int, c_void_p, c_long, c_int,
                                          c_void_p]),
    "savit_cast_bf16": (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    "savit_hwcn_to_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "savit_normalize_to_nhwc_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float, ctypes.POINTER(c_float),
                                             ctypes.POINTER(c_float), c_void_p]),
    "savit_batch_mixup_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p]),
    "savit_batch_cutmix_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "savit_gemm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                               c_int, c_int, c_void_p]),
    "savit_layernorm_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_long, c_long, c_float, c_void_p]),
    "savit_attention_fwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "savit_gemm_f32_ex": (c_int, [POINTER(GemmF32Args), c_void_p]),
    "savit_softmax_rows_f32": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_void_p]),
    "savit_softmax_rows_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p]),
    "savit_head_mix_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_long, c_void_p]),
    "savit_layernorm_bwd_f32": (c_int, [c_void_p] * 7 + [c_int, c_int, c_long, c_long, c_float, c_void_p]),
    "savit_colsum_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "savit_layerscale_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "savit_softmax_xent_grad_f32": (c_int, [c_void_p, c_void_p, c_float, c_float, c_void_p, c_int, c_int, c_void_p]),
    "savit_patchify_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "savit_patchify_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "savit_assemble_tokens_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "savit_timer_create": (c_int, [c_int, POINTER(c_void_p)]),
    "savit_timer_record": (c_int, [c_void_p, c_int, c_void_p]),
    "savit_timer_elapsed_ms": (c_int, [c_void_p, c_int, c_int, POINTER(c_float)]),
    "savit_timer_destroy": (c_int, [c_void_p]),
    "savit_spin": (c_int, [c_long, c_void_p]),
    "savit_hold_cus": (c_int, [c_int, c_long, c_void_p]),
    "savit_set_cu_budget": (c_int, [c_int]),
    "savit_zero_bytes": (c_int, [c_void_p, c_long, c_void_p]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def load() -> ctypes.CDLL:
    """Load libsavit.so (once).  Raises RuntimeError loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c \"import __graft_entry__ as g; g.build()\"` "
            "(or `make -C self-attention-experiments-vision_amd/csrc`). There is no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise RuntimeError(f"libsavit.so does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.savit_abi_version() != ABI_VERSION:
        raise RuntimeError("libsavit.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code == 0:
        return
    if code == SAVIT_EINVAL:
        raise ValueError(f"{what}: argument contract violated (SAVIT_EINVAL)")
    raise RuntimeError(f"{what}: HIP error {code}")
