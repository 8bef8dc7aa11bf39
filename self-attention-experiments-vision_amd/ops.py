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
                  dcolsum: Optional[torch.Tensor] = None, round_params: bool = True, workspace: Optional[torch.Tensor] = None,
                  extra_slab: Optional[torch.Tensor] = None, extra_out: Optional[torch.Tensor] = None):
    """Backward of layernorm_fwd; dgamma/dbeta/dcolsum are accumulated (caller zeroes).  workspace: optional
    uint8/any CUDA tensor of >= savit_layernorm_bwd_workspace_bytes(rows, d) bytes (allocated here if None).
    extra_slab [rows_x, n_x] fp32 + extra_out [n_x] fp32: the finalize launch also adds the slab's column sums to extra_out
    (savit_layernorm_bwd_ex: the bias-gradient partials of the GELU' GEMM; rows wider than 64 columns only)."""
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
    for t in (dgamma, dbeta, dcolsum, gamma):
        if t is not None and t.numel() != d:
            raise ValueError("per-column buffers must have d elements")
    if mean.numel() < rows or rstd.numel() < rows:
        raise ValueError("mean/rstd too small")
    L = _lib.load()
    need = L.savit_layernorm_bwd_workspace_bytes(rows, d)
    if workspace is None:
        workspace = torch.empty((max(need, 16),), dtype=torch.uint8, device=x.device)
    wbytes = workspace.numel() * workspace.element_size()
    if not workspace.is_cuda or wbytes < need:
        raise ValueError("layernorm_bwd: workspace too small")
    if extra_slab is not None or extra_out is not None:
        _chk(extra_slab, f32, "extra_slab", 2)
        _chk(extra_out, f32, "extra_out", 1)
        if not extra_slab.is_contiguous() or extra_out.numel() != extra_slab.shape[1]:
            raise ValueError("extra_slab must be contiguous [rows, n] and extra_out [n]")
        _lib.check(L.savit_layernorm_bwd_ex(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres_in), _p(dx), _p(dx_bf16), _p(dgamma),
                                            _p(dbeta), _p(dcolsum), rows, d, xs, os_, int(round_params), _p(workspace), wbytes,
                                            _p(extra_slab), extra_slab.shape[0], extra_slab.shape[1], _p(extra_out), _stream()),
                   "savit_layernorm_bwd_ex")
        return dx
    _lib.check(L.savit_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres_in), _p(dx), _p(dx_bf16), _p(dgamma),
                                     _p(dbeta), _p(dcolsum), rows, d, xs, os_, int(round_params), _p(workspace), wbytes, _stream()),
               "savit_layernorm_bwd")
    return dx


# ---------------------------------------------------------------------------------------------
def gemm_tn(A: torch.Tensor, Bt: torch.Tensor, C: torch.Tensor, epilogue: int, *, C2=None, bias=None, aux=None, colscale=None,
            rowscale=None, colsum=None, alpha: float = 1.0, alpha_cols: int = 0, rows_per_sample: int = 1,
            round_out_bf16: bool = False, round_bias_bf16: bool = True, tile: int = 0, M: Optional[int] = None,
            patch_geom: Optional[Tuple[int, int, int, int]] = None, cu_budget: int = 0):
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
    for name, t, n in (("bias", bias, N), ("colscale", colscale, N)):
        if t is not None:
            _chk(t, f32, name, 1)
            if t.numel() < n:
                raise ValueError(f"{name} too small")
    colsum_rows = 0
    if colsum is not None:
        if colsum.dim() == 2:  # [rows, N] slab of per-row-tile partial sums (no atomics); rows from gemm_colsum_rows()
            _chk(colsum, f32, "colsum", 2)
            colsum_rows = colsum.shape[0]
            if colsum.shape[1] != N or not colsum.is_contiguous() or colsum_rows != gemm_colsum_rows(Mv, N, K, tile):
                raise ValueError("colsum slab must be contiguous [gemm_colsum_rows(M, N, K, tile), N]")
        else:
            _chk(colsum, f32, "colsum", 1)
            if colsum.numel() < N:
                raise ValueError("colsum too small")
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
    a.colsum_rows = colsum_rows
    a.cu_budget = int(cu_budget)  # CUs the launch may count on (0 = all): tile choice, and the grid of the persistent 320 x 256 kernel
    L = _lib.load()
    _lib.check(L.savit_gemm_bf16_tn(ctypes.byref(a), _stream()), "savit_gemm_bf16_tn")
    return C


def gemm_colsum_rows(M: int, N: int, K: int, tile: int = 0) -> int:
    """Height of the partial-sum slab `gemm_tn(..., colsum=<2-D>)` writes for this shape (one row per row tile and wave row)."""
    r = _lib.load().savit_gemm_colsum_rows(int(M), int(N), int(K), int(tile))
    if r < 0:
        raise ValueError("unknown tile id")
    return r


def colsum_finalize(slab: torch.Tensor, out: torch.Tensor, accumulate: bool = True) -> torch.Tensor:
    """out[n] (+)= sum_r slab[r][n]"""
    _chk(slab, f32, "slab", 2)
    _chk(out, f32, "out", 1)
    if not slab.is_contiguous() or out.numel() < slab.shape[1]:
        raise ValueError("colsum_finalize: contiguous slab [rows, N], out [>= N]")
    L = _lib.load()
    _lib.check(L.savit_colsum_finalize(_p(slab), slab.shape[0], slab.shape[1], _p(out), int(accumulate), _stream()), "savit_colsum_finalize")
    return out


def gemm_wgrad(X: torch.Tensor, dY: torch.Tensor, dW: torch.Tensor, splits: int = 0, M: Optional[int] = None,
               patch_geom: Optional[Tuple[int, int, int, int]] = None, workspace: Optional[torch.Tensor] = None):
    """dW[Kin,Nout] += X[M,Kin]^T @ dY[M,Nout].  patch_geom as in gemm_tn (X = images).  workspace (uint8, from
    wgrad_workspace): the splits of the reduction go through partial slabs + an ordered sum (bitwise reproducible); without it they
    are added with fp32 atomics."""
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
    if workspace is not None:
        if workspace.dtype != torch.uint8 or not workspace.is_cuda or not workspace.is_contiguous():
            raise ValueError("workspace must be a contiguous uint8 GPU tensor")
        _lib.check(L.savit_gemm_bf16_wgrad_ws(_p(X), _p(dY), _p(dW), Mv, Kin, Nout, ldx, lddy, lddw, int(splits), pg[0], pg[1], pg[2],
                                              pg[3], _p(workspace), workspace.numel(), _stream()), "savit_gemm_bf16_wgrad_ws")
        return dW
    _lib.check(L.savit_gemm_bf16_wgrad(_p(X), _p(dY), _p(dW), Mv, Kin, Nout, ldx, lddy, lddw, int(splits), pg[0], pg[1], pg[2],
                                       pg[3], _stream()), "savit_gemm_bf16_wgrad")
    return dW


def gemm_wgrad_grouped(problems, tile: int = 256, overwrite: bool = False, sumsq32: Optional[torch.Tensor] = None):
    """dW_i += X_i^T @ dY_i for every (X_i, dY_i, dW_i) of `problems` in ONE launch, one workgroup per tile x tile output tile, each
    reducing over all rows (no split, no slabs; savit_gemm_bf16_wgrad_grouped).  At most 64 problems.
    overwrite: dW_i = X_i^T @ dY_i instead (first touch: dW is not read).  sumsq32: fp32 [32] accumulators that receive the sum of
    squares of everything the launch stored (savit_gemm_bf16_wgrad_grouped_ex)."""
    arr = (_lib.WgradProblem * len(problems))()
    if sumsq32 is not None:
        _chk(sumsq32, f32, "sumsq32", 1)
        if sumsq32.numel() < 32:
            raise ValueError("sumsq32 needs 32 accumulators")
    for q, pr in zip(arr, problems):
        q.overwrite = int(bool(overwrite))
        X, dY, dW = pr[:3]
        if len(pr) == 5:  # (X, dY, dW, tile_begin, tile_count): a range of the weight's output tiles
            q.tile_begin, q.tile_count = int(pr[3]), int(pr[4])
        _chk(X, bf16, "X", 2)
        _chk(dY, bf16, "dY", 2)
        _chk(dW, f32, "dW", 2)
        Kin, Nout, lddw = _rows2d(dW, "dW")
        Mv, kx, ldx = _rows2d(X, "X")
        ry, cy, lddy = _rows2d(dY, "dY")
        if kx < Kin or cy < Nout or ry < Mv:
            raise ValueError("operand shapes do not match dW")
        q.X, q.dY, q.dW, q.M, q.Kin, q.Nout, q.ldx, q.lddy, q.lddw = _p(X), _p(dY), _p(dW), Mv, Kin, Nout, ldx, lddy, lddw
    _lib.check(_lib.load().savit_gemm_bf16_wgrad_grouped_ex(arr, len(problems), int(tile), _p(sumsq32), _stream()), "savit_gemm_bf16_wgrad_grouped")


def wgrad_workspace(M: int, Kin: int, Nout: int, splits: int = 0, patch: int = 0, device="cuda") -> torch.Tensor:
    """Scratch for gemm_wgrad(..., workspace=...): one partial [Kin, Nout] fp32 slab per split of the reduction over M."""
    n = int(_lib.load().savit_gemm_wgrad_workspace_bytes(int(M), int(Kin), int(Nout), int(splits), int(patch)))
    return torch.empty(max(n, 16), dtype=torch.uint8, device=device)


# ---------------------------------------------------------------------------------------------
def attention_fwd(qkv: torch.Tensor, B: int, N: int, H: int, out: Optional[torch.Tensor] = None,
                  lse: Optional[torch.Tensor] = None, head_dim: int = 64):
    """Fused softmax attention (attention.py:39-58).  qkv bf16 [B*N, 3*H*head_dim] with pre-scaled queries."""
    _chk(qkv, bf16, "qkv", 2)
    rows, cols, ld = _rows2d(qkv, "qkv")
    d = H * head_dim
    if rows < B * N or cols < 3 * d:
        raise ValueError(f"qkv {tuple(qkv.shape)} too small for B={B} N={N} H={H}")
    o = out if out is not None else torch.empty((B * N, d), dtype=bf16, device=qkv.device)
    lse = lse if lse is not None else torch.empty((B, H, N), dtype=f32, device=qkv.device)
    _chk(o, bf16, "out", 2)
    _chk(lse, f32, "lse")
    if not o.is_contiguous() or tuple(o.shape) != (B * N, d) or lse.numel() < B * H * N or not lse.is_contiguous():
        raise ValueError("attention_fwd: bad output buffers")
    L = _lib.load()
    _lib.check(L.savit_attention_fwd(_p(qkv), _p(o), _p(lse), B, N, H, head_dim, ld, _stream()), "savit_attention_fwd")
    return o, lse


def attention_bwd(qkv: torch.Tensor, o: torch.Tensor, d_o: torch.Tensor, lse: torch.Tensor, B: int, N: int, H: int,
                  dq_scale: float, dqkv: Optional[torch.Tensor] = None, head_dim: int = 64):
    _chk(qkv, bf16, "qkv", 2)
    _chk(o, bf16, "o", 2)
    _chk(d_o, bf16, "d_o", 2)
    _chk(lse, f32, "lse")
    rows, cols, ld = _rows2d(qkv, "qkv")
    d = H * head_dim
    if rows < B * N or cols < 3 * d:
        raise ValueError("qkv too small")
    for name, t in (("o", o), ("d_o", d_o)):
        if tuple(t.shape) != (B * N, d) or not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous [B*N, d]")
    if lse.numel() < B * H * N or not lse.is_contiguous():
        raise ValueError("lse too small")
    if dqkv is None:
        dqkv = torch.empty((B * N, 3 * d), dtype=bf16, device=qkv.device)
    _chk(dqkv, bf16, "dqkv", 2)
    r2, c2, ld2 = _rows2d(dqkv, "dqkv")
    if r2 < B * N or c2 < 3 * d or ld2 != ld:
        raise ValueError("dqkv must have qkv's shape and row stride")
    L = _lib.load()
    _lib.check(L.savit_attention_bwd(_p(qkv), _p(o), _p(d_o), _p(lse), _p(dqkv), B, N, H, head_dim, ld, float(dq_scale), _stream()),
               "savit_attention_bwd")
    return dqkv


# ---------------------------------------------------------------------------------------------
def cls_pos_rows(cls: torch.Tensor, pos: torch.Tensor, x0: torch.Tensor, B: int, N: int):
    """x0[b, 0, :] = cls + pos[0]; x0 fp32 [B*N, d]."""
    _chk(x0, f32, "x0", 2)
    d = x0.shape[1]
    _chk(cls, f32, "cls")
    _chk(pos, f32, "pos")
    if cls.numel() != d or pos.numel() < d or x0.shape[0] < B * N or not x0.is_contiguous():
        raise ValueError("cls_pos_rows: bad shapes")
    L = _lib.load()
    _lib.check(L.savit_cls_pos_rows(_p(cls), _p(pos), _p(x0), B, N * d, d, _stream()), "savit_cls_pos_rows")


def pos_cls_grad(dx0: torch.Tensor, dpos: torch.Tensor, dcls: Optional[torch.Tensor], B: int, N: int):
    _chk(dx0, f32, "dx0", 2)
    d = dx0.shape[1]
    _chk(dpos, f32, "dpos")
    if dx0.shape[0] < B * N or not dx0.is_contiguous() or dpos.numel() != N * d or not dpos.is_contiguous():
        raise ValueError("pos_cls_grad: bad shapes")
    if dcls is not None:
        _chk(dcls, f32, "dcls")
        if dcls.numel() != d:
            raise ValueError("dcls size")
    L = _lib.load()
    _lib.check(L.savit_pos_cls_grad(_p(dx0), _p(dpos), _p(dcls), B, N, d, int(dcls is not None), _stream()), "savit_pos_cls_grad")


def softmax_xent(logits: torch.Tensor, labels: torch.Tensor, label_smoothing: float = 0.1, grad_scale: Optional[float] = None,
                 mix_labels: Optional[torch.Tensor] = None, ratio: Optional[torch.Tensor] = None, loss_rows=None, loss_mean=None,
                 dlogits=None, dbias=None, top1=None, top5=None):
    """train.py:83-90.  logits fp32 [B, C]; labels int32 [B]."""
    _chk(logits, f32, "logits", 2)
    B, C, ld = _rows2d(logits, "logits")
    _chk(labels, torch.int32, "labels", 1)
    if labels.numel() != B:
        raise ValueError("labels size")
    if (mix_labels is None) != (ratio is None):
        raise ValueError("mix_labels and ratio go together")
    if mix_labels is not None:
        _chk(mix_labels, torch.int32, "mix_labels", 1)
        _chk(ratio, f32, "ratio", 1)
        if mix_labels.numel() != B or ratio.numel() != B:
            raise ValueError("mix sizes")
    ld_dz = 0
    if dlogits is not None:
        _chk(dlogits, bf16, "dlogits", 2)
        r, c, ld_dz = _rows2d(dlogits, "dlogits")
        if r < B or c < C:
            raise ValueError("dlogits too small")
    for name, t, n in (("loss_rows", loss_rows, B), ("loss_mean", loss_mean, 1), ("dbias", dbias, C), ("top1", top1, B), ("top5", top5, B)):
        if t is not None:
            _chk(t, f32, name)
            if t.numel() < n:
                raise ValueError(f"{name} too small")
    gs = (1.0 / B) if grad_scale is None else float(grad_scale)
    L = _lib.load()
    _lib.check(L.savit_softmax_xent(_p(logits), ld, _p(labels), _p(mix_labels), _p(ratio), float(label_smoothing), gs, _p(loss_rows),
                                    _p(loss_mean), _p(dlogits), ld_dz, _p(dbias), _p(top1), _p(top5), B, C, _stream()), "savit_softmax_xent")


def sumsq(g: torch.Tensor, out: torch.Tensor):
    _chk(g, f32, "g", 1)
    _chk(out, f32, "out")
    L = _lib.load()
    _lib.check(L.savit_sumsq(_p(g), g.numel(), _p(out), _stream()), "savit_sumsq")


def _range_array(g: torch.Tensor, ranges):
    import ctypes

    flat = []
    for off, n in ranges:
        if off < 0 or n < 0 or off % 4 or n % 4 or off + n > g.numel():
            raise ValueError("ranges must be multiples of 4 floats inside the buffer")
        flat += [int(off), int(n)]
    return (ctypes.c_long * len(flat))(*flat)


def zero_ranges(g: torch.Tensor, ranges):
    """g[off : off + n] = 0 for every (off, n) of `ranges` in one launch (savit_zero_ranges)."""
    _chk(g, f32, "g", 1)
    _lib.check(_lib.load().savit_zero_ranges(_p(g), _range_array(g, ranges), len(ranges), _stream()), "savit_zero_ranges")


def sumsq_ranges(g: torch.Tensor, ranges, out: torch.Tensor, slots: Optional[torch.Tensor] = None):
    """out[0] += sum of g^2 over the ranges (+ the partial sums in `slots`) (savit_sumsq_ranges)."""
    _chk(g, f32, "g", 1)
    _chk(out, f32, "out")
    if slots is not None:
        _chk(slots, f32, "slots", 1)
    _lib.check(_lib.load().savit_sumsq_ranges(_p(g), _range_array(g, ranges), len(ranges), _p(slots), slots.numel() if slots is not None else 0,
                                              _p(out), _stream()), "savit_sumsq_ranges")


def layernorm_bwd_finalize_jobs(jobs):
    """jobs = [(workspace, rows, d, nf, (out0, out1, out2, out3), (extra_slab, extra_out) or None)]: reduce the column-sum slabs that
    deferred layernorm_bwd / layerscale_bwd calls (every output pointer None) left in their workspaces - one launch
    (savit_layernorm_bwd_finalize_jobs)."""
    L = _lib.load()
    arr = (_lib.ColsumJob * len(jobs))()
    for q, (ws, rows, d, nf, outs, extra) in zip(arr, jobs):
        q.partial, q.nblk, q.d, q.nf = _p(ws), int(L.savit_layernorm_bwd_grid(int(rows))), int(d), int(nf)
        for i, o in enumerate(outs):
            if o is not None:
                _chk(o, f32, "out", 1)
                q.out[i] = _p(o)
        if extra is not None:
            slab, xout = extra
            _chk(slab, f32, "extra_slab", 2)
            _chk(xout, f32, "extra_out", 1)
            q.extra_slab, q.extra_rows, q.extra_n, q.extra_out = _p(slab), slab.shape[0], slab.shape[1], _p(xout)
    _lib.check(L.savit_layernorm_bwd_finalize_jobs(arr, len(jobs), _stream()), "savit_layernorm_bwd_finalize_jobs")


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float, b1: float = 0.9,
               b2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0, grad_sumsq: Optional[torch.Tensor] = None,
               max_norm: float = 0.0, grad_scale: float = 1.0, mirror: Optional[torch.Tensor] = None):
    """mirror: optional bf16 tensor of p's size that receives bf16(updated p) (savit_adamw_step_mirror)."""
    for name, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _chk(t, f32, name, 1)
        if t.numel() != p.numel() or not t.is_contiguous():
            raise ValueError("adamw_step: size mismatch")
    if grad_sumsq is not None:
        _chk(grad_sumsq, f32, "grad_sumsq")
    if mirror is not None:
        _chk(mirror, bf16, "mirror", 1)
        if mirror.numel() != p.numel() or not mirror.is_contiguous():
            raise ValueError("adamw_step: mirror size mismatch")
    L = _lib.load()
    _lib.check(L.savit_adamw_step_mirror(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(b1), float(b2), float(eps),
                                         float(weight_decay), int(step), _p(grad_sumsq), float(max_norm), float(grad_scale), _p(mirror),
                                         _stream()), "savit_adamw_step_mirror")


def cast_transpose_bf16(src: torch.Tensor, batch: int, R: int, C: int, src_bs: int, dst_n: Optional[torch.Tensor], dn_bs: int,
                        dst_t: Optional[torch.Tensor], dt_bs: int, ld_n: Optional[int] = None, ld_t: Optional[int] = None):
    """src fp32 (flat) holds `batch` [R,C] matrices src_bs elements apart -> bf16 copies as-is / transposed."""
    _chk(src, f32, "src")
    if src.numel() < (batch - 1) * src_bs + R * C:
        raise ValueError("src too small")
    ld_n = C if ld_n is None else ld_n
    ld_t = R if ld_t is None else ld_t
    if dst_n is not None:
        _chk(dst_n, bf16, "dst_n")
        if dst_n.numel() < (batch - 1) * dn_bs + (R - 1) * ld_n + C:
            raise ValueError("dst_n too small")
    if dst_t is not None:
        _chk(dst_t, bf16, "dst_t")
        if dst_t.numel() < (batch - 1) * dt_bs + (C - 1) * ld_t + R:
            raise ValueError("dst_t too small")
    L = _lib.load()
    _lib.check(L.savit_cast_transpose_bf16(_p(src), src_bs, batch, R, C, _p(dst_n), dn_bs, ld_n, _p(dst_t), dt_bs, ld_t, _stream()),
               "savit_cast_transpose_bf16")


def cast_bf16(src: torch.Tensor, dst: torch.Tensor):
    _chk(src, f32, "src")
    _chk(dst, bf16, "dst")
    if src.numel() != dst.numel() or not src.is_contiguous() or not dst.is_contiguous():
        raise ValueError("cast_bf16: size/contiguity")
    L = _lib.load()
    _lib.check(L.savit_cast_bf16(_p(src), _p(dst), src.numel(), _stream()), "savit_cast_bf16")
    return dst


def hwcn_to_nhwc_bf16(src: torch.Tensor, dst: Optional[torch.Tensor] = None):
    """[H,W,C,N] fp32 -> [N,H,W,C] bf16  (train.py:80-81)."""
    _chk(src, f32, "src", 4)
    if not src.is_contiguous():
        raise ValueError("src must be contiguous")
    H, W, C, N = src.shape
    if dst is None:
        dst = torch.empty((N, H, W, C), dtype=bf16, device=src.device)
    _chk(dst, bf16, "dst", 4)
    if tuple(dst.shape) != (N, H, W, C) or not dst.is_contiguous():
        raise ValueError("dst shape")
    L = _lib.load()
    _lib.check(L.savit_hwcn_to_nhwc_bf16(_p(src), _p(dst), H, W, C, N, _stream()), "savit_hwcn_to_nhwc_bf16")
    return dst


# ---------------------------------------------------------------------------------------------
def th_attention_fwd(qkv: torch.Tensor, T1: torch.Tensor, T2: torch.Tensor, B: int, N: int, H: int, head_dim: int = 48):
    """Talking-heads attention forward (CaiT).  Returns (o bf16 [B*N, d], s_buf, p_buf) - the last two feed backward."""
    _chk(qkv, bf16, "qkv", 2)
    _chk(T1, f32, "T1", 2)
    _chk(T2, f32, "T2", 2)
    rows, cols, ld = _rows2d(qkv, "qkv")
    d = H * head_dim
    if rows < B * N or cols < 3 * d or tuple(T1.shape) != (H, H) or tuple(T2.shape) != (H, H) or not T1.is_contiguous() or not T2.is_contiguous():
        raise ValueError("th_attention_fwd: bad shapes")
    Np = (N + 7) // 8 * 8
    s_buf = torch.empty((B, H, N, Np), dtype=bf16, device=qkv.device)
    p_buf = torch.empty((B, H, N, Np), dtype=bf16, device=qkv.device)
    o = torch.empty((B * N, d), dtype=bf16, device=qkv.device)
    L = _lib.load()
    _lib.check(L.savit_th_attention_fwd(_p(qkv), _p(T1), _p(T2), _p(s_buf), _p(p_buf), _p(o), B, N, H, head_dim, ld, Np, _stream()),
               "savit_th_attention_fwd")
    return o, s_buf, p_buf


def th_attention_bwd(qkv, T1, T2, s_buf, p_buf, d_o, dT1, dT2, B: int, N: int, H: int, dq_scale: float, head_dim: int = 48):
    """Backward of th_attention_fwd; p_buf is overwritten.  dT1/dT2 accumulate.  Returns dqkv bf16 [B*N, ld]."""
    for name, t, dt in (("qkv", qkv, bf16), ("s_buf", s_buf, bf16), ("p_buf", p_buf, bf16), ("d_o", d_o, bf16), ("T1", T1, f32), ("T2", T2, f32),
                        ("dT1", dT1, f32), ("dT2", dT2, f32)):
        _chk(t, dt, name)
    rows, cols, ld = _rows2d(qkv, "qkv")
    d = H * head_dim
    Np = s_buf.shape[-1]
    if tuple(s_buf.shape) != (B, H, N, Np) or tuple(p_buf.shape) != (B, H, N, Np) or tuple(d_o.shape) != (B * N, d) or not d_o.is_contiguous():
        raise ValueError("th_attention_bwd: bad shapes")
    if dT1.numel() != H * H or dT2.numel() != H * H:
        raise ValueError("dT sizes")
    ds_buf = torch.empty_like(s_buf)
    dqkv = torch.empty((rows, cols), dtype=bf16, device=qkv.device)
    L = _lib.load()
    need = L.savit_th_attention_bwd_workspace_bytes(B, N, H)
    ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=qkv.device)
    _lib.check(L.savit_th_attention_bwd(_p(qkv), _p(T1), _p(T2), _p(s_buf), _p(p_buf), _p(d_o), _p(ds_buf), _p(dqkv), _p(dT1), _p(dT2), B, N, H,
                                        head_dim, ld, Np, float(dq_scale), _p(ws), ws.numel(), _stream()), "savit_th_attention_bwd")
    return dqkv


def th_fused_supported(N: int, H: int, head_dim: int) -> bool:
    """True when the fused talking-heads kernels (S / P' in LDS) cover this geometry."""
    return bool(_lib.load().savit_th_fused_supported(N, H, head_dim))


def th_fused_attention_fwd(qkv: torch.Tensor, T1: torch.Tensor, T2: torch.Tensor, B: int, N: int, H: int, head_dim: int = 48,
                           workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused talking-heads attention forward (csrc/th_fused.hip): returns o bf16 [B*N, d]; nothing is kept for backward."""
    _chk(qkv, bf16, "qkv", 2)
    _chk(T1, f32, "T1", 2)
    _chk(T2, f32, "T2", 2)
    rows, cols, ld = _rows2d(qkv, "qkv")
    d = H * head_dim
    if rows < B * N or cols < 3 * d or tuple(T1.shape) != (H, H) or tuple(T2.shape) != (H, H) or not T1.is_contiguous() or not T2.is_contiguous():
        raise ValueError("th_fused_attention_fwd: bad shapes")
    L = _lib.load()
    if not L.savit_th_fused_supported(N, H, head_dim):
        raise ValueError(f"th_fused_attention_fwd: geometry N={N} H={H} head_dim={head_dim} is not covered (use th_attention_fwd)")
    need = L.savit_th_fused_fwd_workspace_bytes(B, N, H, head_dim)
    if workspace is None:
        workspace = torch.empty((max(need, 16),), dtype=torch.uint8, device=qkv.device)
    o = torch.empty((B * N, d), dtype=bf16, device=qkv.device)
    _lib.check(L.savit_th_fused_attention_fwd(_p(qkv), _p(T1), _p(T2), _p(o), B, N, H, head_dim, ld, _p(workspace), workspace.numel(), _stream()),
               "savit_th_fused_attention_fwd")
    return o


def th_fused_attention_bwd(qkv, T1, T2, d_o, dT1, dT2, B: int, N: int, H: int, dq_scale: float, head_dim: int = 48):
    """Backward of th_fused_attention_fwd from QKV and dO alone.  dT1/dT2 accumulate.  Returns dqkv bf16 [B*N, ld]."""
    for name, t, dt in (("qkv", qkv, bf16), ("d_o", d_o, bf16), ("T1", T1, f32), ("T2", T2, f32), ("dT1", dT1, f32), ("dT2", dT2, f32)):
        _chk(t, dt, name)
    rows, cols, ld = _rows2d(qkv, "qkv")
    d = H * head_dim
    if tuple(d_o.shape) != (B * N, d) or not d_o.is_contiguous() or dT1.numel() != H * H or dT2.numel() != H * H:
        raise ValueError("th_fused_attention_bwd: bad shapes")
    L = _lib.load()
    Np = (N + 7) // 8 * 8
    p_buf = torch.empty((B, H, N, Np), dtype=bf16, device=qkv.device)
    ds_buf = torch.empty_like(p_buf)
    dqkv = torch.empty((rows, cols), dtype=bf16, device=qkv.device)
    need = L.savit_th_fused_bwd_workspace_bytes(B, N, H, head_dim)
    ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=qkv.device)
    _lib.check(L.savit_th_fused_attention_bwd(_p(qkv), _p(T1), _p(T2), _p(d_o), _p(p_buf), _p(ds_buf), _p(dqkv), _p(dT1), _p(dT2), B, N, H, head_dim,
                                              ld, Np, float(dq_scale), _p(ws), ws.numel(), _stream()), "savit_th_fused_attention_bwd")
    return dqkv


# --------------------------------------------------------------------------------------------- input path (row f-2)
IMAGENET_1K_MEAN, IMAGENET_1K_STD = (0.475, 0.452, 0.398), (0.232, 0.228, 0.229)      # data/constants.py:7-8
IMAGENET_21K_MEAN, IMAGENET_21K_STD = (0.494, 0.473, 0.415), (0.228, 0.224, 0.230)    # data/constants.py:9-10


def normalize_to_nhwc_bf16(src: torch.Tensor, mean=IMAGENET_1K_MEAN, std=IMAGENET_1K_STD, scale: Optional[float] = None,
                           layout: str = "NHWC", dst: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(src*scale - mean)/std -> bf16 [N,H,W,C]  (preprocess.py:176-179 + train.py:80-81).  src: fp32 in the loader's
    [H,W,C,N] layout (layout='HWCN'), or fp32 / uint8 [N,H,W,C].  scale defaults to 1/255 for uint8 and 1 for floats."""
    if not src.is_cuda or src.dim() != 4 or not src.is_contiguous():
        raise ValueError("src must be a contiguous 4-D GPU tensor")
    if layout == "HWCN":
        if src.dtype != f32:
            raise ValueError("the [H,W,C,N] loader layout is fp32")
        H, W, C, N = src.shape
        fmt = 0
    elif layout == "NHWC":
        N, H, W, C = src.shape
        if src.dtype == f32:
            fmt = 1
        elif src.dtype == torch.uint8:
            fmt = 2
        else:
            raise ValueError("NHWC source must be fp32 or uint8")
    else:
        raise ValueError("layout must be 'NHWC' or 'HWCN'")
    if len(mean) != C or len(std) != C or C > 4:
        raise ValueError("mean/std must have one entry per channel (C <= 4)")
    if scale is None:
        scale = 1.0 / 255.0 if src.dtype == torch.uint8 else 1.0
    if dst is None:
        dst = torch.empty((N, H, W, C), dtype=bf16, device=src.device)
    _chk(dst, bf16, "dst", 4)
    if tuple(dst.shape) != (N, H, W, C) or not dst.is_contiguous():
        raise ValueError("dst shape")
    import ctypes as _ct
    m = (_ct.c_float * C)(*[float(v) for v in mean])
    s = (_ct.c_float * C)(*[float(v) for v in std])
    L = _lib.load()
    _lib.check(L.savit_normalize_to_nhwc_bf16(_p(src), fmt, _p(dst), H, W, C, N, float(scale), m, s, _stream()), "savit_normalize_to_nhwc_bf16")
    return dst


def batch_mixup(x: torch.Tensor, weight: torch.Tensor, index: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x[b]*weight[b] + x[index[b]]*(1-weight[b])  (augment_ops.py:176-179); x bf16 [B,...], weight fp32 [B], index int32 [B]."""
    _chk(x, bf16, "x")
    _chk(weight, f32, "weight", 1)
    _chk(index, torch.int32, "index", 1)
    B = x.shape[0]
    per = x.numel() // max(B, 1)
    if not x.is_contiguous() or weight.numel() != B or index.numel() != B or per % 8 != 0:
        raise ValueError("batch_mixup: contiguous x, one weight/index per sample, elements per image % 8 == 0")
    if B and (int(index.min()) < 0 or int(index.max()) >= B):
        raise ValueError("index out of range")
    if out is None:
        out = torch.empty_like(x)
    _chk(out, bf16, "out")
    if out.shape != x.shape or out.data_ptr() == x.data_ptr() or not out.is_contiguous():
        raise ValueError("out must be a distinct contiguous tensor of x's shape")
    L = _lib.load()
    _lib.check(L.savit_batch_mixup_bf16(_p(x), _p(out), _p(weight), _p(index), B, per, _stream()), "savit_batch_mixup_bf16")
    return out


def batch_cutmix(x: torch.Tensor, box: torch.Tensor, index: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """where(inside box[b], x[b], x[index[b]])  (augment_ops.py:136-138); x bf16 [B,H,W,C], box int32 [B,4] = y0,y1,x0,x1."""
    _chk(x, bf16, "x", 4)
    _chk(box, torch.int32, "box", 2)
    _chk(index, torch.int32, "index", 1)
    B, H, W, C = x.shape
    if not x.is_contiguous() or tuple(box.shape) != (B, 4) or index.numel() != B or not box.is_contiguous():
        raise ValueError("batch_cutmix: contiguous x [B,H,W,C], box [B,4], index [B]")
    if B and (int(index.min()) < 0 or int(index.max()) >= B):
        raise ValueError("index out of range")
    if out is None:
        out = torch.empty_like(x)
    _chk(out, bf16, "out", 4)
    if out.shape != x.shape or out.data_ptr() == x.data_ptr() or not out.is_contiguous():
        raise ValueError("out must be a distinct contiguous tensor of x's shape")
    L = _lib.load()
    _lib.check(L.savit_batch_cutmix_bf16(_p(x), _p(out), _p(box), _p(index), B, H, W, C, _stream()), "savit_batch_cutmix_bf16")
    return out


# --------------------------------------------------------------------------------------------- MLP-Mixer glue
def transpose_bf16(src: torch.Tensor, dst: Optional[torch.Tensor], R: int, Cc: int, resid: Optional[torch.Tensor] = None,
                   out_f32: Optional[torch.Tensor] = None, round_out_bf16: bool = False, rowsum: Optional[torch.Tensor] = None):
    """Per-image transpose (mlp_mixer.py:19,23): src bf16 [B, >=R, ld_src], of which rows < R and columns < Cc are read;
    dst bf16 [B, >=Cc, ld_dst >= R] receives dst[b, c, r] = src[b, r, c]; out_f32 [B, Cc, R] = resid + that transpose
    (mlp_mixer.py:24), optionally rounded through bf16; rowsum fp32 [R] += sums over images and columns (R % 4 == 0 here: the
    partial sums go through a slab and savit_colsum_finalize)."""
    _chk(src, bf16, "src", 3)
    if not src.is_contiguous() or src.shape[1] < R or src.shape[2] < Cc:
        raise ValueError("transpose_bf16: src must be contiguous [B, >=R, >=Cc]")
    B = src.shape[0]
    ld_dst, dst_bs = R, 0
    if dst is not None:
        _chk(dst, bf16, "dst", 3)
        if not dst.is_contiguous() or dst.shape[0] != B or dst.shape[1] < Cc or dst.shape[2] < R:
            raise ValueError("transpose_bf16: dst must be contiguous [B, >=Cc, >=R]")
        ld_dst, dst_bs = dst.shape[2], dst.shape[1] * dst.shape[2]
    if (resid is None) != (out_f32 is None):
        raise ValueError("transpose_bf16: resid and out_f32 come together")
    if out_f32 is not None:
        _chk(resid, f32, "resid", 3)
        _chk(out_f32, f32, "out_f32", 3)
        if tuple(resid.shape) != (B, Cc, R) or tuple(out_f32.shape) != (B, Cc, R) or not resid.is_contiguous() or not out_f32.is_contiguous():
            raise ValueError("transpose_bf16: resid / out_f32 must be contiguous [B, Cc, R]")
        if dst is not None and ld_dst != R:
            raise ValueError("transpose_bf16: with both outputs the bf16 pitch must equal R")
        ld_dst = R
    if dst is None and out_f32 is None:
        raise ValueError("transpose_bf16: no output")
    L = _lib.load()
    slab = None
    if rowsum is not None:
        _chk(rowsum, f32, "rowsum", 1)
        if rowsum.numel() < R or R % 4:
            raise ValueError("transpose_bf16: rowsum needs >= R elements and R % 4 == 0")
        slab = torch.empty((L.savit_transpose_rowsum_rows(B, Cc), R), dtype=f32, device=src.device)
    _lib.check(L.savit_transpose_bf16(_p(src), src.shape[1] * src.shape[2], src.shape[2], _p(dst), dst_bs, ld_dst, B, R, Cc, _p(resid),
                                      _p(out_f32), int(round_out_bf16), _p(slab), R, _stream()), "savit_transpose_bf16")
    if slab is not None:
        colsum_finalize(slab, rowsum, accumulate=True)
    return dst if dst is not None else out_f32


def token_mean_fwd(h: torch.Tensor) -> torch.Tensor:
    """jnp.mean(x, axis=1) (mlp_mixer.py:62): h bf16 [B, L, d] -> bf16 [B, d], fp32 accumulation."""
    _chk(h, bf16, "h", 3)
    if not h.is_contiguous():
        raise ValueError("token_mean_fwd: h must be contiguous")
    B, Ltok, d = h.shape
    z = torch.empty((B, d), dtype=bf16, device=h.device)
    _lib.check(_lib.load().savit_token_mean_fwd(_p(h), _p(z), B, Ltok, d, _stream()), "savit_token_mean_fwd")
    return z


def token_mean_bwd(dz: torch.Tensor, Ltok: int) -> torch.Tensor:
    """Backward of the token mean: dh[b, l, :] = bf16(dz[b, :] / L)."""
    _chk(dz, bf16, "dz", 2)
    if not dz.is_contiguous():
        raise ValueError("token_mean_bwd: dz must be contiguous")
    B, d = dz.shape
    dh = torch.empty((B, int(Ltok), d), dtype=bf16, device=dz.device)
    _lib.check(_lib.load().savit_token_mean_bwd(_p(dz), _p(dh), B, int(Ltok), d, _stream()), "savit_token_mean_bwd")
    return dh


# --------------------------------------------------------------------------------------------- TNT glue
def tnt_pixel_gather(images: torch.Tensor, patch: int, t: int, ld_out: Optional[int] = None) -> torch.Tensor:
    """PixelEmbedBlock's rearranges (tnt.py:21-29): images bf16 NHWC -> bf16 [B*(S/P)^2*(P/t)^2, ld_out] (zero beyond C*t*t)."""
    _chk(images, bf16, "images", 4)
    if not images.is_contiguous():
        raise ValueError("tnt_pixel_gather: images must be contiguous NHWC")
    B, S, S2, C = images.shape
    if S != S2 or S % patch or patch % t:
        raise ValueError("tnt_pixel_gather: bad geometry")
    F = C * t * t
    ld = ld_out or F
    out = torch.zeros((B * (S // patch) ** 2 * (patch // t) ** 2, ld), dtype=bf16, device=images.device)
    _lib.check(_lib.load().savit_tnt_pixel_gather(_p(images), _p(out), B, S, patch, t, C, ld, _stream()), "savit_tnt_pixel_gather")
    return out


def add_rows_periodic(x: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """x[r, :] += pos[r mod period, :] in place (AddAbsPosEmbed on TNT's pixel stream, tnt.py:170)."""
    _chk(x, f32, "x", 2)
    _chk(pos, f32, "pos", 2)
    if not x.is_contiguous() or not pos.is_contiguous() or x.shape[1] != pos.shape[1]:
        raise ValueError("add_rows_periodic: contiguous [rows, d] and [period, d]")
    _lib.check(_lib.load().savit_add_rows_periodic(_p(x), _p(pos), x.shape[0], pos.shape[0], x.shape[1], _stream()), "savit_add_rows_periodic")
    return x


def tnt_inner2outer_add(patch: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """out[b, 0] = patch[b, 0]; out[b, 1+p] = patch[b, 1+p] + y[b, p]  (tnt.py:49-50).  patch fp32 [B, N, d], y bf16 [B, N-1, d]."""
    _chk(patch, f32, "patch", 3)
    _chk(y, bf16, "y", 3)
    B, N, d = patch.shape
    if tuple(y.shape) != (B, N - 1, d) or not patch.is_contiguous() or not y.is_contiguous():
        raise ValueError("tnt_inner2outer_add: shapes")
    out = torch.empty_like(patch)
    _lib.check(_lib.load().savit_tnt_inner2outer_add(_p(patch), _p(y), _p(out), B, N, d, _stream()), "savit_tnt_inner2outer_add")
    return out


def cast_colsum(src: torch.Tensor, colsum: Optional[torch.Tensor] = None, want_bf16: bool = True):
    """bf16 copy of fp32 [rows, d] and / or colsum[:] += column sums of src."""
    _chk(src, f32, "src", 2)
    if not src.is_contiguous():
        raise ValueError("cast_colsum: src must be contiguous")
    dst = torch.empty(src.shape, dtype=bf16, device=src.device) if want_bf16 else None
    _lib.check(_lib.load().savit_cast_colsum(_p(src), _p(dst), _p(colsum), src.shape[0], src.shape[1], _stream()), "savit_cast_colsum")
    return dst


def tnt_inner2outer_split(dt: torch.Tensor, dres: torch.Tensor, dbias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dres += dt in place; returns dy bf16 [B, N-1, d] = dt[:, 1:]; dbias += column sums of dy."""
    _chk(dt, f32, "dt", 3)
    _chk(dres, f32, "dres", 3)
    B, N, d = dt.shape
    if dres.shape != dt.shape or not dt.is_contiguous() or not dres.is_contiguous():
        raise ValueError("tnt_inner2outer_split: shapes")
    dy = torch.empty((B, N - 1, d), dtype=bf16, device=dt.device)
    _lib.check(_lib.load().savit_tnt_inner2outer_split(_p(dt), _p(dres), _p(dy), _p(dbias), B, N, d, _stream()), "savit_tnt_inner2outer_split")
    return dy


def gather_rows_bf16(src: torch.Tensor, row_stride: int, B: int, d: int) -> torch.Tensor:
    _chk(src, f32, "src")
    dst = torch.empty((B, d), dtype=bf16, device=src.device)
    _lib.check(_lib.load().savit_gather_rows_bf16(_p(src), row_stride, _p(dst), B, d, _stream()), "savit_gather_rows_bf16")
    return dst


def scatter_rows(src: torch.Tensor, dst: torch.Tensor, row_stride: int, dst_bf16: Optional[torch.Tensor] = None):
    _chk(src, bf16, "src", 2)
    _chk(dst, f32, "dst")
    _lib.check(_lib.load().savit_scatter_rows(_p(src), _p(dst), _p(dst_bf16), row_stride, src.shape[0], src.shape[1], _stream()), "savit_scatter_rows")
    return dst


def seq16_attention_fwd(qkv: torch.Tensor, nseq: int) -> torch.Tensor:
    """One wave per 16-token, 4-head (padded to 16 columns) sequence: qkv bf16 [nseq*16, 192] -> o bf16 [nseq*16, 64]."""
    _chk(qkv, bf16, "qkv", 2)
    if tuple(qkv.shape) != (nseq * 16, 192) or not qkv.is_contiguous():
        raise ValueError("seq16_attention_fwd: qkv must be contiguous [nseq*16, 192]")
    o = torch.empty((nseq * 16, 64), dtype=bf16, device=qkv.device)
    _lib.check(_lib.load().savit_seq16_attention_fwd(_p(qkv), _p(o), nseq, 16, 4, 16, 192, _stream()), "savit_seq16_attention_fwd")
    return o


def seq16_attention_bwd(qkv: torch.Tensor, d_o: torch.Tensor, nseq: int, dq_scale: float) -> torch.Tensor:
    _chk(qkv, bf16, "qkv", 2)
    _chk(d_o, bf16, "d_o", 2)
    if tuple(qkv.shape) != (nseq * 16, 192) or tuple(d_o.shape) != (nseq * 16, 64) or not qkv.is_contiguous() or not d_o.is_contiguous():
        raise ValueError("seq16_attention_bwd: shapes")
    dqkv = torch.empty_like(qkv)
    _lib.check(_lib.load().savit_seq16_attention_bwd(_p(qkv), _p(d_o), _p(dqkv), nseq, 16, 4, 16, 192, float(dq_scale), _stream()),
               "savit_seq16_attention_bwd")
    return dqkv
