"""Tensor-level wrappers over the C-ABI kernels.  PyTorch is used only for device memory and streams:
every wrapper validates dtype / device / contiguity on the host (a wrong shape must never reach a
hand-written kernel), then passes raw pointers + the current HIP stream to libsavit.so."""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Tuple

import torch

from . import lib as _lib

bf16 = torch.bfloat16
f32 = torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, name: str, ndim: Optional[int] = None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA (HIP) tensor - there is no CPU path")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got {tuple(t.shape)}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")


def _rows2d(t: torch.Tensor, name: str) -> Tuple[int, int, int]:
    """(rows, cols, row stride in elements) of a 2-D row-major view."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected a 2-D row-major tensor")
    return t.shape[0], t.shape[1], t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# ---------------------------------------------------------------------------------------------
def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-6, round_params: bool = True,
                  out: Optional[torch.Tensor] = None, mean: Optional[torch.Tensor] = None, rstd: Optional[torch.Tensor] = None):
    """flax nn.LayerNorm(dtype=bf16) forward (models/vit.py:19).  x fp32 [rows, d] (row-strided view allowed)."""
    _chk(x, f32, "x", 2)
    _chk(gamma, f32, "gamma", 1)
    _chk(beta, f32, "beta", 1)
    rows, d, xs = _rows2d(x, "x")
    if gamma.numel() != d or beta.numel() != d:
        raise ValueError("gamma/beta size mismatch")
    y = out if out is not None else torch.empty((rows, d), dtype=bf16, device=x.device)
    mean = mean if mean is not None else torch.empty((rows,), dtype=f32, device=x.device)
    rstd = rstd if rstd is not None else torch.empty((rows,), dtype=f32, device=x.device)
    _chk(y, bf16, "out", 2)
    if tuple(y.shape) != (rows, d) or not y.is_contiguous() or mean.numel() < rows or rstd.numel() < rows:
        raise ValueError("layernorm_fwd: bad output buffers")
    L = _lib.load()
    _lib.check(L.savit_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, d, xs, float(eps),
                                     int(round_params), _stream()), "savit_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
                  dgamma: torch.Tensor, dbeta: torch.Tensor, dres_in: Optional[torch.Tensor] = None,
                  dx: Optional[torch.Tensor] = None, dx_bf16: Optional[torch.Tensor] = None,
                  dcolsum: Optional[torch.Tensor] = None, round_params: bool = True):
    """Backward of layernorm_fwd; dgamma/dbeta/dcolsum are accumulated (caller zeroes)."""
    _chk(dy, bf16, "dy", 2)
    _chk(x, f32, "x", 2)
    rows, d, xs = _rows2d(x, "x")
    if tuple(dy.shape) != (rows, d) or not dy.is_contiguous():
        raise ValueError("dy must be contiguous [rows, d]")
    if dx is None:
        dx = torch.empty((rows, d), dtype=f32, device=x.device)
    _chk(dx, f32, "dx", 2)
    _, _, os_ = _rows2d(dx, "dx")
    for name, t, dt in (("dres_in", dres_in, f32), ("dx_bf16", dx_bf16, bf16)):
        if t is not None:
            _chk(t, dt, name, 2)
            r2, d2, s2 = _rows2d(t, name)
            if (r2, d2) != (rows, d) or s2 != os_:
                raise ValueError(f"{name}: must share dx's shape and row stride")
    for name, t in (("dgamma", dgamma), ("dbeta", dbeta), ("dcolsum", dcolsum), ("mean", mean), ("rstd", rstd), ("gamma", gamma)):
        if t is not None:
            _chk(t, f32, name, 1)
    if dgamma.numel() != d or dbeta.numel() != d or (dcolsum is not None and dcolsum.numel() != d) or gamma.numel() != d:
        raise ValueError("per-column buffers must have d elements")
    if mean.numel() < rows or rstd.numel() < rows:
        raise ValueError("mean/rstd too small")
    L = _lib.load()
    _lib.check(L.savit_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres_in), _p(dx), _p(dx_bf16), _p(dgamma),
                                     _p(dbeta), _p(dcolsum), rows, d, xs, os_, int(round_params), _stream()), "savit_layernorm_bwd")
    return dx


# ---------------------------------------------------------------------------------------------
def gemm_tn(A: torch.Tensor, Bt: torch.Tensor, C: torch.Tensor, epilogue: int, *, C2=None, bias=None, aux=None, colscale=None,
            rowscale=None, colsum=None, alpha: float = 1.0, alpha_cols: int = 0, rows_per_sample: int = 1,
            round_out_bf16: bool = False, round_bias_bf16: bool = True, tile: int = 0, M: Optional[int] = None,
            patch_geom: Optional[Tuple[int, int, int, int]] = None):
    """C = epilogue(A @ Bt^T).  A bf16 [M,K], Bt bf16 [N,K] (K contiguous).  See include/savit.h for the
    epilogues.  patch_geom = (img_size, patch, tokens, token_offset) with A = NHWC bf16 images."""
    _chk(Bt, bf16, "Bt", 2)
    N, K, ldb = _rows2d(Bt, "Bt")
    a = _lib.GemmArgs()
    if epilogue == _lib.EPI_PATCH:
        _chk(A, bf16, "images", 4)
        if not A.is_contiguous() or patch_geom is None:
            raise ValueError("patch mode needs contiguous NHWC images and patch_geom")
        img, patch, tokens, tok_off = patch_geom
        B = A.shape[0]
        if tuple(A.shape) != (B, img, img, 3):
            raise ValueError("images must be [B, img, img, 3]")
        Mv = B * (img // patch) ** 2
        lda = 0
        a.img_size, a.patch, a.tokens, a.token_offset = img, patch, tokens, tok_off
        need_rows = B * tokens
    else:
        _chk(A, bf16, "A", 2)
        Mv, Ka, lda = _rows2d(A, "A")
        if Ka != K:
            raise ValueError(f"K mismatch: A has {Ka}, Bt has {K}")
        if M is not None:
            if M > Mv:
                raise ValueError("M exceeds A rows")
            Mv = M
        need_rows = Mv
    cdt = bf16 if epilogue in (_lib.EPI_BF16, _lib.EPI_BIAS_GELU, _lib.EPI_DGELU) else f32
    _chk(C, cdt, "C", 2)
    Cr, Cc, ldc = _rows2d(C, "C")
    if Cr < need_rows or Cc < N:
        raise ValueError(f"C too small: {tuple(C.shape)} for {need_rows}x{N}")
    if C2 is not None:
        _chk(C2, bf16, "C2", 2)
        if tuple(C2.shape) != tuple(C.shape) or C2.stride(0) != C.stride(0):
            raise ValueError("C2 must match C")
    ldaux = 0
    if aux is not None:
        adt = bf16 if epilogue == _lib.EPI_DGELU else f32
        _chk(aux, adt, "aux", 2)
        ar, ac, ldaux = _rows2d(aux, "aux")
        min_rows = (patch_geom[2] if epilogue == _lib.EPI_PATCH else need_rows)
        if ar < min_rows or ac < N:
            raise ValueError("aux too small")
    for name, t, n in (("bias", bias, N), ("colscale", colscale, N), ("colsum", colsum, N)):
        if t is not None:
            _chk(t, f32, name, 1)
            if t.numel() < n:
                raise ValueError(f"{name} too small")
    if rowscale is not None:
        _chk(rowscale, f32, "rowscale", 1)
        if rowscale.numel() * rows_per_sample < Mv:
            raise ValueError("rowscale too small")
    a.A, a.Bt, a.C, a.C2 = _p(A), _p(Bt), _p(C), _p(C2)
    a.bias, a.aux, a.colscale, a.rowscale, a.colsum = _p(bias), _p(aux), _p(colscale), _p(rowscale), _p(colsum)
    a.M, a.N, a.K = Mv, N, K
    a.lda, a.ldb, a.ldc, a.ldaux = lda, ldb, ldc, ldaux
    a.epilogue = epilogue
    a.alpha, a.alpha_cols, a.rows_per_sample = float(alpha), int(alpha_cols), int(rows_per_sample)
    a.round_out_bf16, a.round_bias_bf16, a.tile = int(round_out_bf16), int(round_bias_bf16), int(tile)
    L = _lib.load()
    _lib.check(L.savit_gemm_bf16_tn(ctypes.byref(a), _stream()), "savit_gemm_bf16_tn")
    return C


def gemm_wgrad(X: torch.Tensor, dY: torch.Tensor, dW: torch.Tensor, splits: int = 0, M: Optional[int] = None,
               patch_geom: Optional[Tuple[int, int, int, int]] = None):
    """dW[Kin,Nout] += X[M,Kin]^T @ dY[M,Nout] (fp32 atomics).  patch_geom as in gemm_tn (X = images)."""
    _chk(dY, bf16, "dY", 2)
    _chk(dW, f32, "dW", 2)
    Kin, Nout, lddw = _rows2d(dW, "dW")
    ry, cy, lddy = _rows2d(dY, "dY")
    if cy < Nout:
        raise ValueError("dY has fewer columns than dW")
    if patch_geom is not None:
        _chk(X, bf16, "images", 4)
        img, patch, tokens, tok_off = patch_geom
        B = X.shape[0]
        if tuple(X.shape) != (B, img, img, 3) or not X.is_contiguous():
            raise ValueError("images must be contiguous [B, img, img, 3]")
        Mv = B * (img // patch) ** 2
        if ry < B * tokens:
            raise ValueError("dY too small")
        ldx = 0
        pg = (patch, img, tokens, tok_off)
    else:
        _chk(X, bf16, "X", 2)
        Mv, kx, ldx = _rows2d(X, "X")
        if kx < Kin:
            raise ValueError("X has fewer columns than dW rows")
        if M is not None:
            Mv = min(Mv, M)
        if ry < Mv:
            raise ValueError("dY has fewer rows than X")
        pg = (0, 0, 0, 0)
    L = _lib.load()
    _lib.check(L.savit_gemm_bf16_wgrad(_p(X), _p(dY), _p(dW), Mv, Kin, Nout, ldx, lddy, lddw, int(splits), pg[0], pg[1], pg[2],
                                       pg[3], _stream()), "savit_gemm_bf16_wgrad")
    return dW
