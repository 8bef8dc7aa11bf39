"""fp32 arithmetic mode: what the reference computes when `create_model`'s dtype stays at its default.

The reference's `create_model(model_name, num_classes=1000, dtype=jnp.float32)` defaults to float32
(/root/reference/models/create_model.py:6-8), BASELINE config 1 is ViT-Tiny/16 in fp32 trained by simple_train.py:72-90, and for
CaiT the reference ALWAYS computes in fp32: create_model.py:50-213 do not forward `dtype` and cait.py:147-154 builds the encoder
without it.  With dtype=float32 every Dense, LayerNorm, softmax and GELU computes and returns fp32.

Engines here run that graph on the GPU with the fp32 kernels of csrc/fp32_path.hip (exact fp32-input MFMA products on the fp32
master weights in place - no bf16 anywhere; scores materialised per (image, head), so any sequence length):
  ViTEngineF32   forward, loss, BACKWARD (every parameter gradient) and the AdamW step - a config-1 train step in fp32;
  CaiTEngineF32  forward, loss, BACKWARD and the AdamW step (round 6): talking-heads attention, LayerScale, stochastic depth
                 (explicit keep masks), class attention - the train step of a `cait_*` model in the arithmetic the reference uses;
They share the parameter layouts (ParamLayout / CaiTLayout: flat fp32 buffer, Flax-shaped views) with the bf16 training engines, so
a tree moves between the two unchanged.  These paths are written for exactness and generality (the fp32 MFMA runs at 1/16 of the
bf16 rate); throughput work lives in the bf16 engines."""
from __future__ import annotations

import ctypes
import math
from typing import Dict, List, Optional

import torch

from . import lib as _lib
from .config import ModelConfig
from .engine import ParamLayout, _Plan, _copy_tree

f32 = torch.float32


class _F32Base:
    """Plan helpers shared by the fp32 engines."""

    def _init_common(self, cfg: ModelConfig, batch: int, device: str):
        if not torch.cuda.is_available():
            raise RuntimeError(f"{type(self).__name__} needs a GPU: there is no CPU path")
        self.L = _lib.load()
        self.cfg, self.B, self.dev = cfg, int(batch), torch.device(device)
        self.params = torch.zeros(self.layout.total, dtype=f32, device=self.dev)
        self.grads: Optional[torch.Tensor] = None
        self.w: Dict[str, torch.Tensor] = {}
        self.adam_m = self.adam_v = None
        self.step_count = 0
        self.weights_stale = False
        C = cfg.num_classes
        self.logits = torch.empty(self.B, C, dtype=f32, device=self.dev)
        self.dlogits = torch.empty(self.B, C, dtype=f32, device=self.dev)
        self.labels = torch.zeros(self.B, dtype=torch.int32, device=self.dev)
        self.loss = torch.zeros(1, dtype=f32, device=self.dev)
        self.loss_rows = torch.empty(self.B, dtype=f32, device=self.dev)
        self.top1 = torch.empty(self.B, dtype=f32, device=self.dev)
        self.top5 = torch.empty(self.B, dtype=f32, device=self.dev)
        self.gnorm_sq = torch.zeros(1, dtype=f32, device=self.dev)

    def e(self, *shape) -> torch.Tensor:
        return torch.empty(*shape, dtype=f32, device=self.dev)

    def set_images(self, images: torch.Tensor):
        S = self.cfg.img_size
        if images.is_cuda and tuple(images.shape) == (S, S, 3, self.B):  # the loader's [H, W, C, N] (train.py:80): a permuted device copy
            self.images.copy_(images.permute(3, 0, 1, 2).to(f32))
            return
        if not images.is_cuda or tuple(images.shape) != (self.B, S, S, 3):
            raise ValueError(f"images must be a GPU tensor [B={self.B},{S},{S},3] (NHWC) or [{S},{S},3,B={self.B}]")
        self.images.copy_(images.to(f32))

    # ---- parameters (same tree as the bf16 engine)
    def param_tree(self) -> dict:
        return self.layout.flax_tree(self.params)

    def grad_tree(self) -> dict:
        if self.grads is None:
            self.grads = torch.zeros_like(self.params)
        return self.layout.flax_tree(self.grads)

    def load_params(self, tree: dict):
        _copy_tree(self.param_tree()["params"], tree["params"] if "params" in tree else tree)

    def _off(self, name: str, buf: Optional[torch.Tensor] = None) -> int:
        return (self.params if buf is None else buf).data_ptr() + self.layout.off[name][0] * 4

    def _gemm(self, P: _Plan, label: str, A, W, C, M, N, K, lda, ldw, ldc, **kw):
        """C = epi(A . W) through savit_gemm_f32_ex; kw: bias, aux, ldaux, colscale, rowscale, rows_per_sample, C2, U, transA, transW,
        batch, inner, sA=(outer, inner), sW=(..), sC=(..), alpha, alpha_cols, act, accumulate, aux_row_mod, rowbias."""
        g = _lib.GemmF32Args()
        g.A, g.W, g.C, g.M, g.N, g.K, g.lda, g.ldw, g.ldc = A, W, C, M, N, K, lda, ldw, ldc
        g.bias, g.aux, g.colscale, g.rowscale, g.C2, g.U = (kw.get(k) for k in ("bias", "aux", "colscale", "rowscale", "C2", "U"))
        g.ldaux = kw.get("ldaux", ldc)
        g.transA, g.transW = int(kw.get("transA", 0)), int(kw.get("transW", 0))
        g.batch, g.inner = int(kw.get("batch", 1)), int(kw.get("inner", 1))
        g.sAo, g.sAi = kw.get("sA", (0, 0))
        g.sWo, g.sWi = kw.get("sW", (0, 0))
        g.sCo, g.sCi = kw.get("sC", (0, 0))
        g.alpha, g.alpha_cols = float(kw.get("alpha", 1.0)), int(kw.get("alpha_cols", 0))
        g.act, g.accumulate, g.rows_per_sample = int(kw.get("act", 0)), int(kw.get("accumulate", 0)), int(kw.get("rows_per_sample", 1))
        g.aux_row_mod, g.rowbias = int(kw.get("aux_row_mod", 0)), kw.get("rowbias")
        P.keep.append(g)
        P.add(self.L.savit_gemm_f32_ex, (ctypes.byref(g),), label)

    def _scores(self, P: _Plan, label: str, q_ptr, k_ptr, S, Nq, Nk, H, hd, ldq, ldk, q_img_stride, k_img_stride):
        """S[b, h] = Q[b, h] K[b, h]^T (attention.py:41) for every (image, head): [B * H, Nq, Nk]."""
        self._gemm(P, label, q_ptr, k_ptr, S.data_ptr(), Nq, Nk, hd, ldq, ldk, Nk, transW=1, batch=self.B * H, inner=H,
                   sA=(q_img_stride, hd), sW=(k_img_stride, hd), sC=(H * Nq * Nk, Nq * Nk))

    def optimizer_step(self, lr: float, weight_decay: float = 0.0, max_norm: float = 0.0, b1: float = 0.9, b2: float = 0.999,
                       eps: float = 1e-8, grad_scale: float = 1.0):
        """optax chain of simple_train.py:25-27 / train.py:25-27 (clip, Adam, weight decay, -lr) on the fp32 parameters."""
        if self.grads is None:
            raise RuntimeError("optimizer_step before any loss_backward")
        if self.adam_m is None:
            self.adam_m, self.adam_v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        s = torch.cuda.current_stream().cuda_stream
        self.step_count += 1
        ss = None
        if max_norm and max_norm > 0:
            self.gnorm_sq.zero_()
            _lib.check(self.L.savit_sumsq(self.grads.data_ptr(), self.grads.numel(), self.gnorm_sq.data_ptr(), s), "savit_sumsq")
            ss = self.gnorm_sq.data_ptr()
        _lib.check(self.L.savit_adamw_step(self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(),
                                           self.params.numel(), float(lr), float(b1), float(b2), float(eps), float(weight_decay),
                                           self.step_count, ss, float(max_norm or 0.0), float(grad_scale), s), "savit_adamw_step")

    def refresh_weights(self):  # the fp32 products read the master weights in place
        self.weights_stale = False

    def _wgrad(self, P: _Plan, label, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw=None, **kw):
        """dW[Kin, Nout] += X[Mr, Kin]^T dY[Mr, Nout] (the weight-gradient form of a Dense: X read transposed in place)."""
        self._gemm(P, label, X, dY, dW, Kin, Nout, Mr, ldx, lddy, Nout if lddw is None else lddw, transA=1, accumulate=1, **kw)

    def _dgrad(self, P: _Plan, label, dY, W, dX, Mr, Kin, Nout, lddy, ldw, ldx, **kw):
        """dX[Mr, Kin] = dY[Mr, Nout] W[Kin, Nout]^T (the input-gradient form: the Flax [in, out] kernel read transposed in place)."""
        self._gemm(P, label, dY, W, dX, Mr, Kin, Nout, lddy, ldw, ldx, transW=1, **kw)

    def loss_fn(self, labels: torch.Tensor, label_smoothing: float = 0.1) -> torch.Tensor:
        """train.py:83-90 on the fp32 logits (one-hot, label smoothing, softmax cross-entropy, mean); also fills top1 / top5."""
        self.labels.copy_(labels.to(torch.int32))
        self.loss.zero_()
        _lib.check(self.L.savit_softmax_xent(self.logits.data_ptr(), self.cfg.num_classes, self.labels.data_ptr(), None, None,
                                             float(label_smoothing), 1.0 / self.B, self.loss_rows.data_ptr(), self.loss.data_ptr(),
                                             None, 0, None, self.top1.data_ptr(), self.top5.data_ptr(),
                                             self.B, self.cfg.num_classes, torch.cuda.current_stream().cuda_stream), "savit_softmax_xent")
        return self.loss


class ViTEngineF32(_F32Base):
    def __init__(self, cfg: ModelConfig, batch: int, device: str = "cuda"):
        if cfg.kind != "vit":
            raise NotImplementedError("ViTEngineF32 handles the ViT family (CaiT: CaiTEngineF32)")
        self.layout = ParamLayout(cfg)
        self._init_common(cfg, batch, device)
        d, F, N, NL, H = cfg.embed_dim, cfg.hidden, cfg.seq_len, cfg.num_layers, cfg.num_heads
        self.M = M = self.B * N
        e = self.e
        self.images = e(self.B, cfg.img_size, cfg.img_size, 3)
        self.patches = e(self.B * cfg.n_patches, cfg.patch_dim)
        self.tok = e(self.B * cfg.n_patches, d)
        # activations, saved per layer (a train step differentiates through all of them; simple_train.py:84-85)
        self.x = [e(M, d) for _ in range(NL + 1)]
        self.xmid = [e(M, d) for _ in range(NL)]
        self.h1 = [e(M, d) for _ in range(NL)]
        self.h2 = [e(M, d) for _ in range(NL)]
        self.qkv = [e(M, 3 * d) for _ in range(NL)]
        self.p = [e(self.B * H, N, N) for _ in range(NL)]   # softmax probabilities (attention.py:48)
        self.o = [e(M, d) for _ in range(NL)]
        self.u = [e(M, F) for _ in range(NL)]               # pre-GELU
        self.a = [e(M, F) for _ in range(NL)]
        self.s = e(self.B * H, N, N)                        # scores / dP / dS scratch
        self.zcls = e(self.B, d)
        # backward scratch
        self.dres = e(M, d)
        self.d_a = e(M, F)
        self.d_h = e(M, d)
        self.d_o = e(M, d)
        self.dqkv = e(M, 3 * d)
        self.d_z = e(self.B, d)
        self._fwd: Optional[_Plan] = None
        self._bwd: Optional[_Plan] = None

    def init_params(self, seed: int = 0):
        from .engine import ViTEngine

        ViTEngine.init_params(self, seed)  # reference initialisers; touches only .params / .layout / .cfg / .weights_stale

    def _build_fwd(self) -> _Plan:
        P, L, cfg = _Plan(), self.L, self.cfg
        d, F, C, N, NL, H, B, M, hd = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers, cfg.num_heads, self.B, self.M, cfg.head_dim
        pp = self._off
        P.add(L.savit_patchify_f32, (self.images.data_ptr(), self.patches.data_ptr(), B, cfg.img_size, cfg.patch), "patchify")
        self._gemm(P, "patch_embed", self.patches.data_ptr(), pp("Wpe"), self.tok.data_ptr(), B * cfg.n_patches, d, cfg.patch_dim, cfg.patch_dim, d, d)
        P.add(L.savit_assemble_tokens_f32, (self.tok.data_ptr(), pp("cls"), pp("pos"), self.x[0].data_ptr(), B, N, d), "tokens")
        for l in range(NL):
            x, xm, qkv = self.x[l].data_ptr(), self.xmid[l].data_ptr(), self.qkv[l].data_ptr()
            P.add(L.savit_layernorm_fwd_f32, (x, pp(f"l{l}.ln1_g"), pp(f"l{l}.ln1_b"), self.h1[l].data_ptr(), M, d, d, d, 1e-6), f"l{l}.ln1")
            self._gemm(P, f"l{l}.qkv", self.h1[l].data_ptr(), pp(f"l{l}.Wqkv"), qkv, M, 3 * d, d, d, 3 * d, 3 * d, alpha=1.0 / math.sqrt(hd), alpha_cols=d)
            self._scores(P, f"l{l}.scores", qkv, qkv + 4 * d, self.s, N, N, H, hd, 3 * d, 3 * d, N * 3 * d, N * 3 * d)
            P.add(L.savit_softmax_rows_f32, (self.s.data_ptr(), self.p[l].data_ptr(), B * H * N, N, N), f"l{l}.softmax")
            self._gemm(P, f"l{l}.pv", self.p[l].data_ptr(), qkv + 8 * d, self.o[l].data_ptr(), N, hd, N, N, 3 * d, d, batch=B * H, inner=H,
                       sA=(H * N * N, N * N), sW=(N * 3 * d, hd), sC=(N * d, hd))
            self._gemm(P, f"l{l}.proj", self.o[l].data_ptr(), pp(f"l{l}.Wo"), xm, M, d, d, d, d, d, aux=x, ldaux=d)
            P.add(L.savit_layernorm_fwd_f32, (xm, pp(f"l{l}.ln2_g"), pp(f"l{l}.ln2_b"), self.h2[l].data_ptr(), M, d, d, d, 1e-6), f"l{l}.ln2")
            self._gemm(P, f"l{l}.fc1", self.h2[l].data_ptr(), pp(f"l{l}.W1"), self.a[l].data_ptr(), M, F, d, d, F, F, bias=pp(f"l{l}.b1"), act=1,
                       C2=self.u[l].data_ptr())
            self._gemm(P, f"l{l}.fc2", self.a[l].data_ptr(), pp(f"l{l}.W2"), self.x[l + 1].data_ptr(), M, d, F, F, d, d, bias=pp(f"l{l}.b2"), aux=xm, ldaux=d)
        P.add(L.savit_layernorm_fwd_f32, (self.x[NL].data_ptr(), pp("lnf_g"), pp("lnf_b"), self.zcls.data_ptr(), B, d, N * d, d, 1e-6), "lnf")  # cls rows only
        self._gemm(P, "head", self.zcls.data_ptr(), pp("Wh"), self.logits.data_ptr(), B, C, d, d, C, C, bias=pp("bh"))
        return P

    def _build_bwd(self) -> _Plan:
        """Reverse-mode gradient of the forward plan (jax.value_and_grad at simple_train.py:84-85): every product is the transposed
        form of its forward GEMM, weight gradients accumulate into self.grads."""
        P, L, cfg = _Plan(), self.L, self.cfg
        d, F, C, N, NL, H, B, M, hd = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers, cfg.num_heads, self.B, self.M, cfg.head_dim
        pp = self._off
        gp = lambda n: self._off(n, self.grads)  # noqa: E731
        dres, d_a, d_h, d_o, dqkv, s = (t.data_ptr() for t in (self.dres, self.d_a, self.d_h, self.d_o, self.dqkv, self.s))
        dl = self.dlogits.data_ptr()

        def wgrad(label, X, dY, dW, Mr, Kin, Nout, ldx, lddy):  # dW[Kin, Nout] += X[Mr, Kin]^T dY[Mr, Nout]
            self._gemm(P, label, X, dY, dW, Kin, Nout, Mr, ldx, lddy, Nout, transA=1, accumulate=1)

        def dgrad(label, dY, W, dX, Mr, Kin, Nout, lddy, ldx, **kw):  # dX[Mr, Kin] = dY[Mr, Nout] W[Kin, Nout]^T
            self._gemm(P, label, dY, W, dX, Mr, Kin, Nout, lddy, Nout, ldx, transW=1, **kw)

        # head (vit.py:95-98) and the final LayerNorm on the cls rows (vit.py:57)
        wgrad("head.wgrad", self.zcls.data_ptr(), dl, gp("Wh"), B, d, C, d, C)
        P.add(L.savit_colsum_f32, (dl, gp("bh"), B, C, C), "head.bgrad")
        dgrad("head.dgrad", dl, pp("Wh"), self.d_z.data_ptr(), B, d, C, C, d)
        P.add(L.savit_layernorm_bwd_f32, (self.d_z.data_ptr(), self.x[NL].data_ptr(), pp("lnf_g"), None, dres, gp("lnf_g"), gp("lnf_b"), B, d, N * d, d,
                                          1e-6), "lnf.bwd")
        for l in range(NL - 1, -1, -1):
            qkv = self.qkv[l].data_ptr()
            # x_{l+1} = x_mid + gelu(h2 W1 + b1) W2 + b2   (ff.py:26-33, vit.py:26-31)
            wgrad(f"l{l}.W2.wgrad", self.a[l].data_ptr(), dres, gp(f"l{l}.W2"), M, F, d, F, d)
            P.add(L.savit_colsum_f32, (dres, gp(f"l{l}.b2"), M, d, d), f"l{l}.b2.grad")
            dgrad(f"l{l}.fc2.dgrad", dres, pp(f"l{l}.W2"), d_a, M, F, d, d, F, act=2, U=self.u[l].data_ptr())  # * gelu'(u)
            wgrad(f"l{l}.W1.wgrad", self.h2[l].data_ptr(), d_a, gp(f"l{l}.W1"), M, d, F, d, F)
            P.add(L.savit_colsum_f32, (d_a, gp(f"l{l}.b1"), M, F, F), f"l{l}.b1.grad")
            dgrad(f"l{l}.fc1.dgrad", d_a, pp(f"l{l}.W1"), d_h, M, d, F, F, d)
            P.add(L.savit_layernorm_bwd_f32, (d_h, self.xmid[l].data_ptr(), pp(f"l{l}.ln2_g"), dres, dres, gp(f"l{l}.ln2_g"), gp(f"l{l}.ln2_b"), M, d, d, d,
                                              1e-6), f"l{l}.ln2.bwd")
            # x_mid = x_l + attn(LN1(x_l)) Wo   (attention.py:21-67, vit.py:19-24)
            wgrad(f"l{l}.Wo.wgrad", self.o[l].data_ptr(), dres, gp(f"l{l}.Wo"), M, d, d, d, d)
            dgrad(f"l{l}.proj.dgrad", dres, pp(f"l{l}.Wo"), d_o, M, d, d, d, d)
            bh = dict(batch=B * H, inner=H)
            # dP = dO V^T ; dV = P^T dO ; dS = P (dP - sum dP P) ; dQ = dS K / sqrt(hd) ; dK = dS^T Q
            self._gemm(P, f"l{l}.dP", d_o, qkv + 8 * d, s, N, N, hd, d, 3 * d, N, transW=1, sA=(N * d, hd), sW=(N * 3 * d, hd), sC=(H * N * N, N * N), **bh)
            self._gemm(P, f"l{l}.dV", self.p[l].data_ptr(), d_o, dqkv + 8 * d, N, hd, N, N, d, 3 * d, transA=1, sA=(H * N * N, N * N), sW=(N * d, hd),
                       sC=(N * 3 * d, hd), **bh)
            P.add(L.savit_softmax_rows_bwd_f32, (self.p[l].data_ptr(), s, s, B * H * N, N, N), f"l{l}.softmax.bwd")
            self._gemm(P, f"l{l}.dQ", s, qkv + 4 * d, dqkv, N, hd, N, N, 3 * d, 3 * d, sA=(H * N * N, N * N), sW=(N * 3 * d, hd), sC=(N * 3 * d, hd),
                       alpha=1.0 / math.sqrt(hd), alpha_cols=hd, **bh)
            self._gemm(P, f"l{l}.dK", s, qkv, dqkv + 4 * d, N, hd, N, N, 3 * d, 3 * d, transA=1, sA=(H * N * N, N * N), sW=(N * 3 * d, hd),
                       sC=(N * 3 * d, hd), **bh)
            wgrad(f"l{l}.Wqkv.wgrad", self.h1[l].data_ptr(), dqkv, gp(f"l{l}.Wqkv"), M, d, 3 * d, d, 3 * d)
            dgrad(f"l{l}.qkv.dgrad", dqkv, pp(f"l{l}.Wqkv"), d_h, M, d, 3 * d, 3 * d, d)
            P.add(L.savit_layernorm_bwd_f32, (d_h, self.x[l].data_ptr(), pp(f"l{l}.ln1_g"), dres, dres, gp(f"l{l}.ln1_g"), gp(f"l{l}.ln1_b"), M, d, d, d,
                                              1e-6), f"l{l}.ln1.bwd")
        # embeddings: dpos, dcls (vit.py:81-85, position_embed.py:56), dWpe per image (patch_embed.py:23-25: token rows 1.. of each image)
        P.add(L.savit_pos_cls_grad, (dres, gp("pos"), gp("cls"), B, N, d, 1), "pos_cls.grad")
        P.add(_gather_patch_rows, (self,), "dtok.gather")  # the patch rows of dx0 (token 0 of each image is cls), contiguous
        wgrad("Wpe.wgrad", self.patches.data_ptr(), self.tok.data_ptr(), gp("Wpe"), B * cfg.n_patches, cfg.patch_dim, d, cfg.patch_dim, d)
        return P

    def forward(self, images: Optional[torch.Tensor] = None) -> torch.Tensor:
        if images is not None:
            self.set_images(images)
        if self._fwd is None:
            self._fwd = self._build_fwd()
        self._fwd.run(torch.cuda.current_stream().cuda_stream)
        return self.logits

    def loss_backward(self, labels: torch.Tensor, label_smoothing: float = 0.1, zero_grads: bool = True) -> torch.Tensor:
        """Loss (train.py:83-90 / simple_train.py:76-83) + the full backward pass into self.grads (fp32)."""
        if self.grads is None:
            self.grads = torch.zeros_like(self.params)
        elif zero_grads:
            self.grads.zero_()
        s = torch.cuda.current_stream().cuda_stream
        self.loss_fn(labels, label_smoothing)
        _lib.check(self.L.savit_softmax_xent_grad_f32(self.logits.data_ptr(), self.labels.data_ptr(), float(label_smoothing), 1.0 / self.B,
                                                      self.dlogits.data_ptr(), self.B, self.cfg.num_classes, s), "savit_softmax_xent_grad_f32")
        self.dres.zero_()
        if self._bwd is None:
            self._bwd = self._build_bwd()
        self._bwd.run(s)
        return self.loss

class CaiTEngineF32(_F32Base):
    """cait.py:140-183 in fp32 - the arithmetic the reference ALWAYS uses for CaiT (its create_model branch ignores dtype):
    forward, loss, and (round 6) the BACKWARD of all of it plus the AdamW step, i.e. the train step `train.py:77-100` would run on a
    `cait_*` model.  Forward-only use keeps one set of activation buffers; the first `loss_backward` switches the engine to saving
    every layer's activations (S and P per (image, head) included: the reference's XLA graph saves them too) and re-runs the forward
    once."""

    SA_SAVED = ("x", "h1", "qkv", "S", "P", "o", "br1", "xmid", "h2", "u", "a", "br2")
    CA_SAVED = ("clsin", "xc", "hc", "qc", "kvc", "pc", "oc", "cbr1", "clsmid", "hq", "uc", "ac", "cbr2")

    def __init__(self, cfg: ModelConfig, batch: int, device: str = "cuda"):
        if cfg.kind != "cait":
            raise ValueError("CaiTEngineF32 needs a CaiT config")
        if cfg.num_heads > 16:
            raise NotImplementedError("savit_head_mix_f32 handles up to 16 heads")
        from .cait_engine import CaiTLayout

        self.layout = CaiTLayout(cfg)
        self._init_common(cfg, batch, device)
        d, F, N, H = cfg.embed_dim, cfg.hidden, cfg.n_patches, cfg.num_heads
        B = self.B
        self.M, self.Mc = B * N, B * (N + 1)
        e = self.e
        self.images = e(B, cfg.img_size, cfg.img_size, 3)
        self.patches = e(self.M, cfg.patch_dim)
        self.x, self.xmid, self.h = e(self.M, d), e(self.M, d), e(self.M, d)
        self.qkv, self.o, self.a = e(self.M, 3 * d), e(self.M, d), e(self.M, F)
        self.s, self.s2 = e(B * H, N, N), e(B * H, N, N)
        # class-attention stage: [cls ; x] rows, cls stream, its 1 x (N + 1) scores
        self.xc, self.hc = e(self.Mc, d), e(self.Mc, d)
        self.cls, self.clsmid, self.hq = e(B, d), e(B, d), e(B, d)
        self.qc, self.kvc, self.oc, self.ac = e(B, d), e(self.Mc, 2 * d), e(B, d), e(B, F)
        self.sc, self.pc = e(B * H, 1, N + 1), e(B * H, 1, N + 1)
        self.zcls = e(B, d)
        self.keep: Optional[torch.Tensor] = None  # [(L + Lc), 2, B] keep masks / keep_prob of a training-mode forward, or None
        self.gen = torch.Generator(device=self.dev)
        self._plans: Dict[tuple, _Plan] = {}
        self.save_activations = False   # set by the first loss_backward: forward then writes per-layer buffers (self.sv)
        self.sv: Optional[Dict[str, List[torch.Tensor]]] = None
        self._saved_training: Optional[bool] = None  # mode of the forward whose activations self.sv holds
        self._bwd_plans: Dict[bool, _Plan] = {}

    def init_params(self, seed: int = 0):
        from .cait_engine import CaiTEngine

        CaiTEngine.init_params(self, seed)

    # ---- per-layer activation buffers of a train step
    def _alloc_saved(self):
        cfg, B, e = self.cfg, self.B, self.e
        d, F, N, NL, NC, H = cfg.embed_dim, cfg.hidden, cfg.n_patches, cfg.num_layers, cfg.num_layers_token_only, cfg.num_heads
        M, Mc, Nk = self.M, self.Mc, N + 1
        shp = {"x": (M, d), "h1": (M, d), "qkv": (M, 3 * d), "S": (B * H, N, N), "P": (B * H, N, N), "o": (M, d), "br1": (M, d), "xmid": (M, d),
               "h2": (M, d), "u": (M, F), "a": (M, F), "br2": (M, d)}
        cshp = {"clsin": (B, d), "xc": (Mc, d), "hc": (Mc, d), "qc": (B, d), "kvc": (Mc, 2 * d), "pc": (B * H, 1, Nk), "oc": (B, d), "cbr1": (B, d),
                "clsmid": (B, d), "hq": (B, d), "uc": (B, F), "ac": (B, F), "cbr2": (B, d)}
        self.sv = {k: [e(*v) for _ in range(NL)] for k, v in shp.items()}
        self.sv.update({k: [e(*v) for _ in range(NC)] for k, v in cshp.items()})
        self.sv["xsa"] = [e(M, d)]  # the SA stage's output (frozen through the class-attention stage, cait.py:157-173)
        # backward scratch
        self.bw = {"dres": e(M, d), "d_br": e(M, d), "d_a": e(M, F), "d_h": e(M, d), "d_o": e(M, d), "dqkv": e(M, 3 * d), "sA": e(B * H, N, N),
                   "sB": e(B * H, N, N), "dT": e(B, H, H), "Tt": e(NL, 2, H, H), "dcls": e(B, d), "d_cbr": e(B, d), "d_ac": e(B, F), "d_hq": e(B, d),
                   "d_oc": e(B, d), "dpc": e(B * H, 1, Nk), "dqc": e(B, d), "dkvc": e(Mc, 2 * d), "d_hc": e(Mc, d), "dxc": e(Mc, d), "d_z": e(B, d)}

    def _build(self, training: bool, save: bool = False) -> _Plan:
        P, L, cfg = _Plan(), self.L, self.cfg
        d, F, C, N, NL, NC, H, B, M, Mc, hd = (cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.n_patches, cfg.num_layers, cfg.num_layers_token_only,
                                              cfg.num_heads, self.B, self.M, self.Mc, cfg.head_dim)
        pp = self._off
        sd = training and cfg.stoch_depth_rate > 0
        sv = self.sv if save else None
        ptr = lambda t: t.data_ptr()  # noqa: E731

        def rs(block, which):  # stochastic depth as the per-image row scale of the residual epilogue (stochastic_depth.py:16-27)
            return dict(rowscale=self.keep[block, which].data_ptr()) if sd else {}

        x0 = ptr(sv["x"][0]) if save else ptr(self.x)
        P.add(L.savit_patchify_f32, (self.images.data_ptr(), self.patches.data_ptr(), B, cfg.img_size, cfg.patch), "patchify")
        self._gemm(P, "patch_embed", self.patches.data_ptr(), pp("Wpe"), x0, M, d, cfg.patch_dim, cfg.patch_dim, d, d, aux=pp("pos"), ldaux=d,
                   aux_row_mod=N)  # + pos_embed, shared by the images (cait.py:143-145, position_embed.py:56)
        x = x0
        for l in range(NL):
            if save:
                h1, qkv, S, Pm, o, xm, h2, a = (ptr(sv[k][l]) for k in ("h1", "qkv", "S", "P", "o", "xmid", "h2", "a"))
                s2, xn = ptr(self.s2), (ptr(sv["x"][l + 1]) if l + 1 < NL else ptr(sv["xsa"][0]))
                extra1, extra_u, extra2 = dict(C2=ptr(sv["br1"][l])), dict(C2=ptr(sv["u"][l])), dict(C2=ptr(sv["br2"][l]))
            else:
                h1 = h2 = ptr(self.h)
                qkv, S, Pm, o, xm, a, s2, xn = ptr(self.qkv), ptr(self.s), ptr(self.s2), ptr(self.o), ptr(self.xmid), ptr(self.a), ptr(self.s), x
                extra1 = extra_u = extra2 = {}
            P.add(L.savit_layernorm_fwd_f32, (x, pp(f"l{l}.ln1_g"), pp(f"l{l}.ln1_b"), h1, M, d, d, d, 1e-6), f"l{l}.ln1")
            self._gemm(P, f"l{l}.qkv", h1, pp(f"l{l}.Wqkv"), qkv, M, 3 * d, d, d, 3 * d, 3 * d, alpha=1.0 / math.sqrt(hd), alpha_cols=d)
            self._gemm(P, f"l{l}.scores", qkv, qkv + 4 * d, S, N, N, hd, 3 * d, 3 * d, N, transW=1, batch=B * H, inner=H, sA=(N * 3 * d, hd),
                       sW=(N * 3 * d, hd), sC=(H * N * N, N * N))                                    # attention.py:41
            P.add(L.savit_head_mix_f32, (pp(f"l{l}.T1"), S, Pm, B, H, N * N), f"l{l}.th1")          # attention.py:44-46  (S' into the P buffer)
            P.add(L.savit_softmax_rows_f32, (Pm, Pm, B * H * N, N, N), f"l{l}.softmax")               # :48  (in place: P)
            P.add(L.savit_head_mix_f32, (pp(f"l{l}.T2"), Pm, s2, B, H, N * N), f"l{l}.th2")         # :50-52  (P')
            self._gemm(P, f"l{l}.pv", s2, qkv + 8 * d, o, N, hd, N, N, 3 * d, d, batch=B * H, inner=H, sA=(H * N * N, N * N), sW=(N * 3 * d, hd),
                       sC=(N * d, hd))
            self._gemm(P, f"l{l}.proj", o, pp(f"l{l}.Wo"), xm, M, d, d, d, d, d, aux=x, ldaux=d, colscale=pp(f"l{l}.ls1"), rows_per_sample=N, **rs(l, 0),
                       **extra1)
            P.add(L.savit_layernorm_fwd_f32, (xm, pp(f"l{l}.ln2_g"), pp(f"l{l}.ln2_b"), h2, M, d, d, d, 1e-6), f"l{l}.ln2")
            self._gemm(P, f"l{l}.fc1", h2, pp(f"l{l}.W1"), a, M, F, d, d, F, F, bias=pp(f"l{l}.b1"), act=1, **extra_u)
            self._gemm(P, f"l{l}.fc2", a, pp(f"l{l}.W2"), xn, M, d, F, F, d, d, bias=pp(f"l{l}.b2"), aux=xm, ldaux=d, colscale=pp(f"l{l}.ls2"),
                       rows_per_sample=N, **rs(l, 1), **extra2)
            x = xn
        x_final_t = (sv["xsa"][0] if NL > 0 else sv["x"][0]) if save else self.x  # the SA stage's output, as a tensor (row copies below)
        # class-attention stage (cait.py:157-173): cls starts as the parameter; x is frozen
        Nk = N + 1
        for c in range(NC):
            if save:
                xc, hc, qc, kvc, pc, oc, clsm, hq, ac = (ptr(sv[k][c]) for k in ("xc", "hc", "qc", "kvc", "pc", "oc", "clsmid", "hq", "ac"))
                cls_in = ptr(sv["clsin"][c])
                cls_out = ptr(sv["clsin"][c + 1]) if c + 1 < NC else ptr(self.cls)
                extra1, extra_u, extra2 = dict(C2=ptr(sv["cbr1"][c])), dict(C2=ptr(sv["uc"][c])), dict(C2=ptr(sv["cbr2"][c]))
            else:
                xc, hc, qc, kvc, pc, oc, clsm, hq, ac = (ptr(t) for t in (self.xc, self.hc, self.qc, self.kvc, self.pc, self.oc, self.clsmid, self.hq, self.ac))
                cls_in = cls_out = ptr(self.cls)
                extra1 = extra_u = extra2 = {}
            sc = ptr(self.sc)
            P.add(_concat_rows, (self, c, save, x_final_t), f"c{c}.concat")  # xc = [cls ; x] (cait.py:98): device-side row copies
            P.add(L.savit_layernorm_fwd_f32, (xc, pp(f"c{c}.ln1_g"), pp(f"c{c}.ln1_b"), hc, Mc, d, d, d, 1e-6), f"c{c}.ln1")
            # q from row 0 only (cait.py:13-15), k and v from all N + 1 rows
            self._gemm(P, f"c{c}.q", hc, pp(f"c{c}.Wqkv"), qc, B, d, d, Nk * d, 3 * d, d, alpha=1.0 / math.sqrt(hd), alpha_cols=d)
            self._gemm(P, f"c{c}.kv", hc, pp(f"c{c}.Wqkv") + 4 * d, kvc, Mc, 2 * d, d, d, 3 * d, 2 * d)
            self._gemm(P, f"c{c}.scores", qc, kvc, sc, 1, Nk, hd, d, 2 * d, Nk, transW=1, batch=B * H, inner=H, sA=(d, hd), sW=(Nk * 2 * d, hd),
                       sC=(H * Nk, Nk))
            P.add(L.savit_softmax_rows_f32, (sc, pc, B * H, Nk, Nk), f"c{c}.softmax")
            self._gemm(P, f"c{c}.pv", pc, kvc + 4 * d, oc, 1, hd, Nk, Nk, 2 * d, d, batch=B * H, inner=H, sA=(H * Nk, Nk), sW=(Nk * 2 * d, hd), sC=(d, hd))
            self._gemm(P, f"c{c}.proj", oc, pp(f"c{c}.Wo"), clsm, B, d, d, d, d, d, aux=cls_in, ldaux=d, colscale=pp(f"c{c}.ls1"), rows_per_sample=1,
                       **rs(NL + c, 0), **extra1)
            P.add(L.savit_layernorm_fwd_f32, (clsm, pp(f"c{c}.ln2_g"), pp(f"c{c}.ln2_b"), hq, B, d, d, d, 1e-6), f"c{c}.ln2")
            self._gemm(P, f"c{c}.fc1", hq, pp(f"c{c}.W1"), ac, B, F, d, d, F, F, bias=pp(f"c{c}.b1"), act=1, **extra_u)
            self._gemm(P, f"c{c}.fc2", ac, pp(f"c{c}.W2"), cls_out, B, d, F, F, d, d, bias=pp(f"c{c}.b2"), aux=clsm, ldaux=d, colscale=pp(f"c{c}.ls2"),
                       rows_per_sample=1, **rs(NL + c, 1), **extra2)
        # final LayerNorm over [cls ; x], of which only row 0 reaches the head (cait.py:175-182)
        P.add(L.savit_layernorm_fwd_f32, (ptr(self.cls), pp("lnf_g"), pp("lnf_b"), self.zcls.data_ptr(), B, d, d, d, 1e-6), "lnf")
        self._gemm(P, "head", self.zcls.data_ptr(), pp("Wh"), self.logits.data_ptr(), B, C, d, d, C, C, bias=pp("bh"))
        return P

    def _build_bwd(self, training: bool) -> _Plan:
        """Reverse-mode gradient of `_build(training, save=True)` (jax.value_and_grad of train.py:94-95 on a CaiT): the class-attention
        stage, the talking-heads SA stage, the embeddings.  Every product is the transposed form of its forward GEMM; the talking-heads
        mixes transpose their H x H matrices; LayerScale x stochastic depth through savit_layerscale_bwd_f32."""
        P, L, cfg, sv, bw = _Plan(), self.L, self.cfg, self.sv, self.bw
        d, F, C, N, NL, NC, H, B, M, Mc, hd = (cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.n_patches, cfg.num_layers, cfg.num_layers_token_only,
                                              cfg.num_heads, self.B, self.M, self.Mc, cfg.head_dim)
        pp = self._off
        gp = lambda n: self._off(n, self.grads)  # noqa: E731
        ptr = lambda t: t.data_ptr()  # noqa: E731
        sd = training and cfg.stoch_depth_rate > 0
        rsp = lambda block, which: (self.keep[block, which].data_ptr() if sd else None)  # noqa: E731
        alpha = 1.0 / math.sqrt(hd)
        Nk, E = N + 1, N * N
        dres, d_br, d_a, d_h, d_o, dqkv, sA, sB, dT = (ptr(bw[k]) for k in ("dres", "d_br", "d_a", "d_h", "d_o", "dqkv", "sA", "sB", "dT"))
        dcls, d_cbr, d_ac, d_hq, d_oc, dpc, dqc, dkvc, d_hc, dxc, d_z = (ptr(bw[k]) for k in ("dcls", "d_cbr", "d_ac", "d_hq", "d_oc", "dpc", "dqc", "dkvc",
                                                                                            "d_hc", "dxc", "d_z"))
        dl = self.dlogits.data_ptr()
        ln_bwd, colsum = L.savit_layernorm_bwd_f32, L.savit_colsum_f32

        def ls_bwd(label, dres_p, branch, ls_name, rs_ptr, rps, out, rows):
            P.add(L.savit_layerscale_bwd_f32, (dres_p, branch, pp(ls_name), rs_ptr, rps, out, gp(ls_name), rows, d), label)

        # ---- head (cait.py:178-182) and the final LayerNorm, whose row 0 alone is read (cait.py:175-178)
        self._wgrad(P, "head.wgrad", self.zcls.data_ptr(), dl, gp("Wh"), B, d, C, d, C)
        P.add(colsum, (dl, gp("bh"), B, C, C), "head.bgrad")
        self._dgrad(P, "head.dgrad", dl, pp("Wh"), d_z, B, d, C, C, C, d)
        P.add(ln_bwd, (d_z, ptr(self.cls), pp("lnf_g"), None, dcls, gp("lnf_g"), gp("lnf_b"), B, d, d, d, 1e-6), "lnf.bwd")
        # ---- class-attention stage, last layer first (cait.py:96-122).  dxc accumulates d / d[cls ; x] over the layers: x is the same
        # tensor for all of them; the cls row is handed to and from the compact cls cotangent by strided device copies.
        P.add(L.savit_zero_bytes, (dxc, Mc * d * 4), "zero.dxc")
        for c in range(NC - 1, -1, -1):
            p = f"c{c}."
            xc, hc, qc, kvc, pc, oc, cbr1, clsm, hq, uc, ac, cbr2 = (ptr(sv[k][c]) for k in ("xc", "hc", "qc", "kvc", "pc", "oc", "cbr1", "clsmid", "hq", "uc",
                                                                                             "ac", "cbr2"))
            # cls_{c+1} = clsmid + ls2 rs2 (gelu(hq W1 + b1) W2 + b2)
            ls_bwd(p + "ls2.bwd", dcls, cbr2, p + "ls2", rsp(NL + c, 1), 1, d_cbr, B)
            self._wgrad(P, p + "W2.wgrad", ac, d_cbr, gp(p + "W2"), B, F, d, F, d)
            P.add(colsum, (d_cbr, gp(p + "b2"), B, d, d), p + "b2.grad")
            self._dgrad(P, p + "fc2.dgrad", d_cbr, pp(p + "W2"), d_ac, B, F, d, d, d, F, act=2, U=uc)
            self._wgrad(P, p + "W1.wgrad", hq, d_ac, gp(p + "W1"), B, d, F, d, F)
            P.add(colsum, (d_ac, gp(p + "b1"), B, F, F), p + "b1.grad")
            self._dgrad(P, p + "fc1.dgrad", d_ac, pp(p + "W1"), d_hq, B, d, F, F, F, d)
            P.add(ln_bwd, (d_hq, clsm, pp(p + "ln2_g"), dcls, dcls, gp(p + "ln2_g"), gp(p + "ln2_b"), B, d, d, d, 1e-6), p + "ln2.bwd")
            # clsmid = cls_c + ls1 rs1 (class-attention(LN1([cls_c ; x])) Wo)
            ls_bwd(p + "ls1.bwd", dcls, cbr1, p + "ls1", rsp(NL + c, 0), 1, d_cbr, B)
            self._wgrad(P, p + "Wo.wgrad", oc, d_cbr, gp(p + "Wo"), B, d, d, d, d)
            self._dgrad(P, p + "proj.dgrad", d_cbr, pp(p + "Wo"), d_oc, B, d, d, d, d, d)
            bh = dict(batch=B * H, inner=H)
            # per (image, head): dp = do V^T ; dV = p^T do ; ds = p (dp - sum dp p) ; dq = ds K / sqrt(hd) ; dK = ds^T q
            self._gemm(P, p + "dP", d_oc, kvc + 4 * d, dpc, 1, Nk, hd, d, 2 * d, Nk, transW=1, sA=(d, hd), sW=(Nk * 2 * d, hd), sC=(H * Nk, Nk), **bh)
            self._gemm(P, p + "dV", pc, d_oc, dkvc + 4 * d, Nk, hd, 1, Nk, d, 2 * d, transA=1, sA=(H * Nk, Nk), sW=(d, hd), sC=(Nk * 2 * d, hd), **bh)
            P.add(L.savit_softmax_rows_bwd_f32, (pc, dpc, dpc, B * H, Nk, Nk), p + "softmax.bwd")
            self._gemm(P, p + "dQ", dpc, kvc, dqc, 1, hd, Nk, Nk, 2 * d, d, sA=(H * Nk, Nk), sW=(Nk * 2 * d, hd), sC=(d, hd), alpha=alpha, alpha_cols=hd, **bh)
            self._gemm(P, p + "dK", dpc, qc, dkvc, Nk, hd, 1, Nk, d, 2 * d, transA=1, sA=(H * Nk, Nk), sW=(d, hd), sC=(Nk * 2 * d, hd), **bh)
            # q from row 0 of every image (pitch Nk * d), k | v from all rows: one fused Wqkv [d, 3 d]
            self._wgrad(P, p + "Wq.wgrad", hc, dqc, gp(p + "Wqkv"), B, d, d, Nk * d, d, 3 * d)
            self._wgrad(P, p + "Wkv.wgrad", hc, dkvc, gp(p + "Wqkv") + 4 * d, Mc, d, 2 * d, d, 2 * d, 3 * d)
            self._dgrad(P, p + "kv.dgrad", dkvc, pp(p + "Wqkv") + 4 * d, d_hc, Mc, d, 2 * d, 2 * d, 3 * d, d)
            self._dgrad(P, p + "q.dgrad", dqc, pp(p + "Wqkv"), d_hc, B, d, d, d, 3 * d, Nk * d, accumulate=1)
            # d[cls_c ; x] += LN1 VJP; the cls row also carries the residual path: put the compact cotangent in before, take the sum out after
            P.add(_cls_rows_copy, (self, "dcls", "dxc", True), p + "dcls.in")
            P.add(ln_bwd, (d_hc, xc, pp(p + "ln1_g"), dxc, dxc, gp(p + "ln1_g"), gp(p + "ln1_b"), Mc, d, d, d, 1e-6), p + "ln1.bwd")
            P.add(_cls_rows_copy, (self, "dcls", "dxc", False), p + "dcls.out")  # (the next layer's dcls.in overwrites row 0 again)
        P.add(colsum, (dcls, gp("cls"), B, d, d), "cls.grad")   # the cls parameter is broadcast over the images (cait.py:157-160)
        P.add(_patch_rows_copy, (self,), "dx.gather")            # dres = d / dx of the SA stage's output: rows 1.. of dxc
        # ---- SA stage, last layer first (cait.py:28-60)
        for l in range(NL - 1, -1, -1):
            p = f"l{l}."
            x, h1, qkv, S, Pm, o, br1, xm, h2, u, a, br2 = (ptr(sv[k][l]) for k in self.SA_SAVED)
            T1t, T2t = ptr(bw["Tt"][l, 0]), ptr(bw["Tt"][l, 1])
            # x_{l+1} = xmid + ls2 rs2 (gelu(h2 W1 + b1) W2 + b2)
            ls_bwd(p + "ls2.bwd", dres, br2, p + "ls2", rsp(l, 1), N, d_br, M)
            self._wgrad(P, p + "W2.wgrad", a, d_br, gp(p + "W2"), M, F, d, F, d)
            P.add(colsum, (d_br, gp(p + "b2"), M, d, d), p + "b2.grad")
            self._dgrad(P, p + "fc2.dgrad", d_br, pp(p + "W2"), d_a, M, F, d, d, d, F, act=2, U=u)
            self._wgrad(P, p + "W1.wgrad", h2, d_a, gp(p + "W1"), M, d, F, d, F)
            P.add(colsum, (d_a, gp(p + "b1"), M, F, F), p + "b1.grad")
            self._dgrad(P, p + "fc1.dgrad", d_a, pp(p + "W1"), d_h, M, d, F, F, F, d)
            P.add(ln_bwd, (d_h, xm, pp(p + "ln2_g"), dres, dres, gp(p + "ln2_g"), gp(p + "ln2_b"), M, d, d, d, 1e-6), p + "ln2.bwd")
            # xmid = x_l + ls1 rs1 (talking-heads attention(LN1(x_l)) Wo)
            ls_bwd(p + "ls1.bwd", dres, br1, p + "ls1", rsp(l, 0), N, d_br, M)
            self._wgrad(P, p + "Wo.wgrad", o, d_br, gp(p + "Wo"), M, d, d, d, d)
            self._dgrad(P, p + "proj.dgrad", d_br, pp(p + "Wo"), d_o, M, d, d, d, d, d)
            bh = dict(batch=B * H, inner=H)
            sAE = (H * E, E)
            # O = P' V: dP' = dO V^T ; dV = P'^T dO with P' = mix(T2, P) recomputed (attention.py:50-57)
            self._gemm(P, p + "dPp", d_o, qkv + 8 * d, sA, N, N, hd, d, 3 * d, N, transW=1, sA=(N * d, hd), sW=(N * 3 * d, hd), sC=sAE, **bh)
            P.add(L.savit_head_mix_f32, (pp(p + "T2"), Pm, sB, B, H, E), p + "th2.re")
            self._gemm(P, p + "dV", sB, d_o, dqkv + 8 * d, N, hd, N, N, d, 3 * d, transA=1, sA=sAE, sW=(N * d, hd), sC=(N * 3 * d, hd), **bh)
            # P'_i = sum_h T2[h, i] P_h (talking_heads.py:13): dT2[h, i] = sum_(b, q, k) P_h dP'_i ; dP_h = sum_i T2[h, i] dP'_i
            self._gemm(P, p + "dT2.part", Pm, sA, dT, H, H, E, E, E, H, transW=1, batch=B, sA=(H * E, 0), sW=(H * E, 0), sC=(H * H, 0))
            P.add(colsum, (dT, gp(p + "T2"), B, H * H, H * H), p + "dT2")
            P.add(L.savit_head_mix_f32, (T2t, sA, sB, B, H, E), p + "th2.bwd")
            P.add(L.savit_softmax_rows_bwd_f32, (Pm, sB, sB, B * H * N, N, N), p + "softmax.bwd")   # dS' (attention.py:48)
            # S'_i = sum_h T1[h, i] S_h: dT1[h, i] = sum S_h dS'_i ; dS_h = sum_i T1[h, i] dS'_i
            self._gemm(P, p + "dT1.part", S, sB, dT, H, H, E, E, E, H, transW=1, batch=B, sA=(H * E, 0), sW=(H * E, 0), sC=(H * H, 0))
            P.add(colsum, (dT, gp(p + "T1"), B, H * H, H * H), p + "dT1")
            P.add(L.savit_head_mix_f32, (T1t, sB, sA, B, H, E), p + "th1.bwd")                      # dS
            # S = (q / sqrt(hd)) k^T (attention.py:39-41; q is stored scaled): dQ = dS K / sqrt(hd) ; dK = dS^T q
            self._gemm(P, p + "dQ", sA, qkv + 4 * d, dqkv, N, hd, N, N, 3 * d, 3 * d, sA=sAE, sW=(N * 3 * d, hd), sC=(N * 3 * d, hd), alpha=alpha,
                       alpha_cols=hd, **bh)
            self._gemm(P, p + "dK", sA, qkv, dqkv + 4 * d, N, hd, N, N, 3 * d, 3 * d, transA=1, sA=sAE, sW=(N * 3 * d, hd), sC=(N * 3 * d, hd), **bh)
            self._wgrad(P, p + "Wqkv.wgrad", h1, dqkv, gp(p + "Wqkv"), M, d, 3 * d, d, 3 * d)
            self._dgrad(P, p + "qkv.dgrad", dqkv, pp(p + "Wqkv"), d_h, M, d, 3 * d, 3 * d, 3 * d, d)
            P.add(ln_bwd, (d_h, x, pp(p + "ln1_g"), dres, dres, gp(p + "ln1_g"), gp(p + "ln1_b"), M, d, d, d, 1e-6), p + "ln1.bwd")
        # ---- embeddings: x0 = patches Wpe + pos (cait.py:143-145): dpos = sum over the images, dWpe = patches^T dx0
        P.add(colsum, (dres, gp("pos"), B, N * d, N * d), "pos.grad")
        self._wgrad(P, "Wpe.wgrad", self.patches.data_ptr(), dres, gp("Wpe"), M, cfg.patch_dim, d, cfg.patch_dim, d)
        return P

    def forward(self, images: Optional[torch.Tensor] = None, is_training: bool = False, keep_masks: Optional[torch.Tensor] = None,
                sd_seed: Optional[int] = None) -> torch.Tensor:
        """keep_masks [(L + Lc), 2, B] of 0 / 1 (tests); drawn from the engine's generator in training mode when absent (sd_seed reseeds
        it first: cait_engine.stochastic_depth_seed, as the bf16 engine)."""
        cfg = self.cfg
        if images is not None:
            self.set_images(images)
        if sd_seed is not None:
            self.gen.manual_seed(int(sd_seed))
        training = bool(is_training) and cfg.stoch_depth_rate > 0
        if training:
            keep = 1.0 - cfg.stoch_depth_rate
            nb = cfg.num_layers + cfg.num_layers_token_only
            if keep_masks is None:
                keep_masks = torch.floor(keep + torch.rand(nb, 2, self.B, device=self.dev, generator=self.gen))
            if self.keep is None:
                self.keep = torch.empty(nb, 2, self.B, dtype=f32, device=self.dev)
            self.keep.copy_(keep_masks.to(device=self.dev, dtype=f32) / keep)
        self._run_forward(training)
        return self.logits

    def _run_forward(self, training: bool):
        save = bool(self.save_activations)
        if save and self.sv is None:
            self._alloc_saved()
        key = (training, save)
        if key not in self._plans:
            if training and self.keep is None:
                raise RuntimeError("keep masks missing")
            self._plans[key] = self._build(training, save)
        # the cls stream starts from the parameter (cait.py:157-160)
        first = self.sv["clsin"][0] if (save and self.cfg.num_layers_token_only > 0) else self.cls
        first.copy_(self.layout.view(self.params, "cls").view(1, -1).expand(self.B, -1))
        self._plans[key].run(torch.cuda.current_stream().cuda_stream)
        self._saved_training = training if save else None
        self._last_training = training

    def loss_backward(self, labels: torch.Tensor, label_smoothing: float = 0.1, zero_grads: bool = True) -> torch.Tensor:
        """Loss (train.py:83-90) + the full backward pass of the LAST forward (same images, same stochastic-depth masks) into
        self.grads (fp32).  The first call switches the engine to saving activations and re-runs that forward once."""
        if self.cfg.num_layers_token_only < 1:
            raise NotImplementedError("CaiT fp32 backward expects at least one class-attention layer (every create_model name has two)")
        if self.grads is None:
            self.grads = torch.zeros_like(self.params)
        elif zero_grads:
            self.grads.zero_()
        training = bool(getattr(self, "_last_training", False))
        if self._saved_training is None or self._saved_training != training:
            self.save_activations = True
            self._run_forward(training)  # (the images and the keep masks of the last forward are still in place)
        s = torch.cuda.current_stream().cuda_stream
        self.loss_fn(labels, label_smoothing)
        _lib.check(self.L.savit_softmax_xent_grad_f32(self.logits.data_ptr(), self.labels.data_ptr(), float(label_smoothing), 1.0 / self.B,
                                                      self.dlogits.data_ptr(), self.B, self.cfg.num_classes, s), "savit_softmax_xent_grad_f32")
        # the transposed talking-heads matrices the two mix VJPs read (strided device copies: memory plumbing, no arithmetic)
        lay = self.layout
        for l in range(self.cfg.num_layers):
            self.bw["Tt"][l, 0].copy_(lay.view(self.params, f"l{l}.T1").t())
            self.bw["Tt"][l, 1].copy_(lay.view(self.params, f"l{l}.T2").t())
        if training not in self._bwd_plans:
            self._bwd_plans[training] = self._build_bwd(training)
        self._bwd_plans[training].run(s)
        return self.loss


class _SavedPlanF32(_F32Base):
    """Shared by the MLP-Mixer and TNT fp32 engines.  Inference keeps one set of activation buffers (`_build(False)`); the first
    `loss_backward` switches the engine to the plan that saves every layer's activations (`_build(True)`) and re-runs that forward
    once (bit-identical logits: the same products in the same order, only the destinations differ)."""

    _plan: Optional[_Plan] = None
    _plan_save: Optional[_Plan] = None
    _bwd: Optional[_Plan] = None
    save_activations = False
    sv: Optional[Dict[str, List[torch.Tensor]]] = None
    _saved = False

    def forward(self, images: Optional[torch.Tensor] = None, is_training: bool = False) -> torch.Tensor:
        if images is not None:
            self.set_images(images)
        self._run_forward()
        return self.logits

    def _run_forward(self):
        save = bool(self.save_activations)
        if save:
            if self.sv is None:
                self._alloc_saved()
            if self._plan_save is None:
                self._plan_save = self._build(True)
            plan = self._plan_save
        else:
            if self._plan is None:
                self._plan = self._build(False)
            plan = self._plan
        plan.run(torch.cuda.current_stream().cuda_stream)
        self._saved = save

    def loss_backward(self, labels: torch.Tensor, label_smoothing: float = 0.1, zero_grads: bool = True) -> torch.Tensor:
        """Loss (train.py:83-90) + the full backward pass of the LAST forward into self.grads (fp32)."""
        if self.grads is None:
            self.grads = torch.zeros_like(self.params)
        elif zero_grads:
            self.grads.zero_()
        if not self._saved:
            self.save_activations = True
            self._run_forward()
        s = torch.cuda.current_stream().cuda_stream
        self.loss_fn(labels, label_smoothing)
        _lib.check(self.L.savit_softmax_xent_grad_f32(self.logits.data_ptr(), self.labels.data_ptr(), float(label_smoothing), 1.0 / self.B,
                                                      self.dlogits.data_ptr(), self.B, self.cfg.num_classes, s), "savit_softmax_xent_grad_f32")
        if self._bwd is None:
            self._bwd = self._build_bwd()
        self._bwd.run(s)
        return self.loss


class MixerEngineF32(_SavedPlanF32):
    """mlp_mixer.py:44-64 in fp32: forward, loss and (round 6) the backward + AdamW step.  Token mixing (MixerBlock :17-24: FFBlock on the
    transposed activation) reads the [n, d] activation of an image as its transposed operand in place and produces the second product
    already transposed back (x[b] += tW2^T a[b]^T, the Dense bias then runs along the rows) - no transposed copy exists, in either
    direction."""

    def __init__(self, cfg: ModelConfig, batch: int, device: str = "cuda"):
        from .mixer_engine import MixerLayout

        self.layout = MixerLayout(cfg)
        self._init_common(cfg, batch, device)
        d, F, n, B = cfg.embed_dim, cfg.hidden, cfg.n_patches, self.B
        self.M = B * n
        e = self.e
        self.images = e(B, cfg.img_size, cfg.img_size, 3)
        self.patches = e(self.M, cfg.patch_dim)
        self.x, self.h, self.a = e(self.M, d), e(self.M, d), e(self.M, F)
        self.at = e(B, d, self.layout.Fp)  # token-mixing hidden activation, [image, channel, Ft]
        self.ones = torch.ones(max(n, d), dtype=f32, device=self.dev)
        self.zmean = e(B, d)

    def init_params(self, seed: int = 0):
        from .mixer_engine import MixerEngine

        MixerEngine.init_params(self, seed)

    def _alloc_saved(self):
        cfg, B, e, lay = self.cfg, self.B, self.e, self.layout
        d, F, n, NL, M = cfg.embed_dim, cfg.hidden, cfg.n_patches, cfg.num_layers, self.M
        Lp, Fp = lay.Lp, lay.Fp
        z = lambda *s_: torch.zeros(*s_, dtype=f32, device=self.dev)  # noqa: E731
        self.sv = {"x": [e(M, d) for _ in range(NL + 1)], "h1": [e(M, d) for _ in range(NL)], "ut": [z(B, d, Fp) for _ in range(NL)],
                   "at": [z(B, d, Fp) for _ in range(NL)], "xmid": [e(M, d) for _ in range(NL)], "h2": [e(M, d) for _ in range(NL)],
                   "u": [e(M, F) for _ in range(NL)], "a": [e(M, F) for _ in range(NL)], "hf": [e(M, d)]}
        # backward scratch; the per-image partials of the token-kernel gradients carry the parameter's padded storage shape (pads stay zero)
        self.bw = {"dres": e(M, d), "d_a": e(M, F), "d_h": e(M, d), "d_ut": z(B, d, Fp), "d_z": e(B, d), "part1": z(B, Lp * Fp), "part2": z(B, Fp * Lp),
                   "rows": z(n * d)}

    def _build(self, save: bool = False) -> _Plan:
        P, L, cfg, lay = _Plan(), self.L, self.cfg, self.layout
        d, F, C, n, NL, B, M = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.n_patches, cfg.num_layers, self.B, self.M
        Ft, Lp, Fp = cfg.tokens_hidden, lay.Lp, lay.Fp
        pp = self._off
        ptr = lambda t: t.data_ptr()  # noqa: E731
        sv = self.sv if save else None
        x = ptr(sv["x"][0]) if save else ptr(self.x)
        P.add(L.savit_patchify_f32, (self.images.data_ptr(), self.patches.data_ptr(), B, cfg.img_size, cfg.patch), "patchify")
        self._gemm(P, "patch_embed", self.patches.data_ptr(), pp("Wpe"), x, M, d, cfg.patch_dim, cfg.patch_dim, d, d, bias=pp("bpe"))
        for l in range(NL):
            if save:
                h1, h2, at, a, xm = (ptr(sv[k][l]) for k in ("h1", "h2", "at", "a", "xmid"))
                xn = ptr(sv["x"][l + 1])
                s_ut, s_u = dict(C2=ptr(sv["ut"][l])), dict(C2=ptr(sv["u"][l]))
                r1, r2 = dict(aux=x, ldaux=d), dict(aux=xm, ldaux=d)          # x_mid = x + ..., x_next = x_mid + ... into fresh buffers
            else:
                h1 = h2 = ptr(self.h)
                at, a, xm, xn = ptr(self.at), ptr(self.a), x, x
                s_ut = s_u = {}
                r1 = r2 = dict(accumulate=1)                                   # in place
            P.add(L.savit_layernorm_fwd_f32, (x, pp(f"l{l}.ln1_g"), pp(f"l{l}.ln1_b"), h1, M, d, d, d, 1e-6), f"l{l}.ln1")
            # a[b] [d, Ft] = gelu(h[b]^T tW1 + tb1)
            self._gemm(P, f"l{l}.tok.fc1", h1, pp(f"l{l}.tW1"), at, d, Ft, n, d, Fp, Fp, transA=1, bias=pp(f"l{l}.tb1"), act=1, batch=B,
                       sA=(n * d, 0), sC=(d * Fp, 0), **s_ut)
            # x[b] [n, d] += tW2^T a[b]^T + tb2 along the rows
            self._gemm(P, f"l{l}.tok.fc2", pp(f"l{l}.tW2"), at, xm, n, d, Ft, Lp, Fp, d, transA=1, transW=1, rowbias=pp(f"l{l}.tb2"),
                       batch=B, sW=(d * Fp, 0), sC=(n * d, 0), **r1)
            P.add(L.savit_layernorm_fwd_f32, (xm, pp(f"l{l}.ln2_g"), pp(f"l{l}.ln2_b"), h2, M, d, d, d, 1e-6), f"l{l}.ln2")
            self._gemm(P, f"l{l}.fc1", h2, pp(f"l{l}.W1"), a, M, F, d, d, F, F, bias=pp(f"l{l}.b1"), act=1, **s_u)
            self._gemm(P, f"l{l}.fc2", a, pp(f"l{l}.W2"), xn, M, d, F, F, d, d, bias=pp(f"l{l}.b2"), **r2)
            x = xn
        hf = ptr(sv["hf"][0]) if save else ptr(self.h)
        P.add(L.savit_layernorm_fwd_f32, (x, pp("lnf_g"), pp("lnf_b"), hf, M, d, d, d, 1e-6), "lnf")
        # mean over the tokens (mlp_mixer.py:61-62) as (1 / n) 1^T z[b]
        self._gemm(P, "token_mean", self.ones.data_ptr(), hf, self.zmean.data_ptr(), 1, d, n, n, d, d, batch=B, sW=(n * d, 0), sC=(d, 0), alpha=1.0 / n,
                   alpha_cols=d)
        self._gemm(P, "head", self.zmean.data_ptr(), pp("Wh"), self.logits.data_ptr(), B, C, d, d, C, C, bias=pp("bh"))
        return P

    def _build_bwd(self) -> _Plan:
        """Reverse-mode gradient of `_build(save=True)`: every product is the transposed form of its forward GEMM; the token-kernel
        gradients, which sum over the images, go through per-image partials (padded like the parameters) and a column sum - two
        images must not add into one output tile at the same time."""
        P, L, cfg, lay, sv, bw = _Plan(), self.L, self.cfg, self.layout, self.sv, self.bw
        d, F, C, n, NL, B, M = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.n_patches, cfg.num_layers, self.B, self.M
        Ft, Lp, Fp = cfg.tokens_hidden, lay.Lp, lay.Fp
        pp = self._off
        gp = lambda nm: self._off(nm, self.grads)  # noqa: E731
        ptr = lambda t: t.data_ptr()  # noqa: E731
        dres, d_a, d_h, d_ut, d_z, part1, part2, rows = (ptr(bw[k]) for k in ("dres", "d_a", "d_h", "d_ut", "d_z", "part1", "part2", "rows"))
        dl, ones = self.dlogits.data_ptr(), self.ones.data_ptr()
        ln_bwd, colsum = L.savit_layernorm_bwd_f32, L.savit_colsum_f32
        # head (mlp_mixer.py:63-64) and the token mean (:61-62): d hf[b, t, :] = d zmean[b] / n for every token
        self._wgrad(P, "head.wgrad", self.zmean.data_ptr(), dl, gp("Wh"), B, d, C, d, C)
        P.add(colsum, (dl, gp("bh"), B, C, C), "head.bgrad")
        self._dgrad(P, "head.dgrad", dl, pp("Wh"), d_z, B, d, C, C, C, d)
        self._gemm(P, "token_mean.bwd", ones, d_z, d_h, n, d, 1, 1, d, d, batch=B, sW=(d, 0), sC=(n * d, 0), alpha=1.0 / n, alpha_cols=d)
        P.add(ln_bwd, (d_h, ptr(sv["x"][NL]), pp("lnf_g"), None, dres, gp("lnf_g"), gp("lnf_b"), M, d, d, d, 1e-6), "lnf.bwd")
        for l in range(NL - 1, -1, -1):
            p = f"l{l}."
            x, h1, ut, at, xm, h2, u, a = (ptr(sv[k][l]) for k in ("x", "h1", "ut", "at", "xmid", "h2", "u", "a"))
            # channel mixing: x_next = x_mid + gelu(h2 W1 + b1) W2 + b2   (mlp_mixer.py:26-30, ff.py:26-33)
            self._wgrad(P, p + "W2.wgrad", a, dres, gp(p + "W2"), M, F, d, F, d)
            P.add(colsum, (dres, gp(p + "b2"), M, d, d), p + "b2.grad")
            self._dgrad(P, p + "fc2.dgrad", dres, pp(p + "W2"), d_a, M, F, d, d, d, F, act=2, U=u)
            self._wgrad(P, p + "W1.wgrad", h2, d_a, gp(p + "W1"), M, d, F, d, F)
            P.add(colsum, (d_a, gp(p + "b1"), M, F, F), p + "b1.grad")
            self._dgrad(P, p + "fc1.dgrad", d_a, pp(p + "W1"), d_h, M, d, F, F, F, d)
            P.add(ln_bwd, (d_h, xm, pp(p + "ln2_g"), dres, dres, gp(p + "ln2_g"), gp(p + "ln2_b"), M, d, d, d, 1e-6), p + "ln2.bwd")
            # token mixing: x_mid[b][t, c] = x[b][t, c] + sum_f tW2[f, t] at[b][c, f] + tb2[t];  at = gelu(ut), ut[b][c, f] = sum_t h1[b][t, c] tW1[t, f] + tb1[f]
            # d tb2[t] = sum over images and channels: first over the images (column sum of [B, n d]), then over the channels of a row
            P.add(L.savit_zero_bytes, (rows, n * d * 4), p + "zero.rows")
            P.add(colsum, (dres, rows, B, n * d, n * d), p + "tb2.part")
            self._gemm(P, p + "tb2.grad", rows, ones, gp(p + "tb2"), n, 1, d, d, 1, 1, accumulate=1)
            # d tW2[f, t] = sum_b sum_c at[b][c, f] dres[b][t, c]: per image into part2 [Fp, Lp], then summed over the images
            self._gemm(P, p + "tW2.part", at, dres, part2, Ft, n, d, Fp, d, Lp, transA=1, transW=1, batch=B, sA=(d * Fp, 0), sW=(n * d, 0), sC=(Fp * Lp, 0))
            P.add(colsum, (part2, gp(p + "tW2"), B, Fp * Lp, Fp * Lp), p + "tW2.grad")
            # d ut[b][c, f] = (sum_t dres[b][t, c] tW2[f, t]) gelu'(ut)
            self._gemm(P, p + "tok.fc2.dgrad", dres, pp(p + "tW2"), d_ut, d, Ft, n, d, Lp, Fp, transA=1, transW=1, act=2, U=ut, batch=B,
                       sA=(n * d, 0), sC=(d * Fp, 0))
            P.add(colsum, (d_ut, gp(p + "tb1"), B * d, Ft, Fp), p + "tb1.grad")
            # d tW1[t, f] = sum_b sum_c h1[b][t, c] d ut[b][c, f]
            self._gemm(P, p + "tW1.part", h1, d_ut, part1, n, Ft, d, d, Fp, Fp, batch=B, sA=(n * d, 0), sW=(d * Fp, 0), sC=(Lp * Fp, 0))
            P.add(colsum, (part1, gp(p + "tW1"), B, Lp * Fp, Lp * Fp), p + "tW1.grad")
            # d h1[b][t, c] = sum_f tW1[t, f] d ut[b][c, f]
            self._gemm(P, p + "tok.fc1.dgrad", pp(p + "tW1"), d_ut, d_h, n, d, Ft, Fp, Fp, d, transW=1, batch=B, sW=(d * Fp, 0), sC=(n * d, 0))
            P.add(ln_bwd, (d_h, x, pp(p + "ln1_g"), dres, dres, gp(p + "ln1_g"), gp(p + "ln1_b"), M, d, d, d, 1e-6), p + "ln1.bwd")
        # patch embedding with bias (mlp_mixer.py:53-55, patch_embed.py:23-25)
        self._wgrad(P, "Wpe.wgrad", self.patches.data_ptr(), dres, gp("Wpe"), M, cfg.patch_dim, d, cfg.patch_dim, d)
        P.add(colsum, (dres, gp("bpe"), M, d, d), "bpe.grad")
        return P


class TNTEngineF32(_SavedPlanF32):
    """tnt.py:150-193 in fp32: the pixel stream [B n npx, di] and the patch stream [B (n + 1), do] (EncoderBlock :66-93): forward, loss
    and (round 6) the backward + AdamW step.  The inner attention kernels are stored head-padded to 16 columns
    (tnt_engine.TNTLayout); the pad columns are zero, so the products over the padded head width are exact - and so are their
    gradients (every cotangent that reaches a pad column is a product with a zero pad)."""

    INNER = ("hi1", "qkvi", "pi", "oi", "ximid", "hi2", "ui", "ai")
    OUTER = ("outer", "ho1", "qkvo", "po", "oo", "xmid", "ho2", "uo", "ao")

    def __init__(self, cfg: ModelConfig, batch: int, device: str = "cuda"):
        from .tnt_engine import HDP, TNTLayout

        self.layout = lay = TNTLayout(cfg)
        self._init_common(cfg, batch, device)
        do, n, npx = cfg.embed_dim, cfg.n_patches, cfg.n_pixels
        B = self.B
        self.HDP = HDP
        self.Mi, self.Ms, self.Mo = B * n * npx, B * n, B * cfg.seq_len
        e = self.e
        self.images = e(B, cfg.img_size, cfg.img_size, 3)
        self.pix = e(self.Mi, lay.pix_in)
        self.patches = e(self.Ms, cfg.patch_dim)
        self.tok = e(self.Ms, do)
        self.act = self._alloc_layer()                               # one set of activations: inference
        self.xi0, self.xo0 = e(self.Mi, cfg.inner_embed_dim), e(self.Mo, do)
        self.si, self.so = e(self.Ms * cfg.inner_num_heads, npx, npx), e(B * cfg.num_heads, cfg.seq_len, cfg.seq_len)  # scores / dP / dS scratch

    def _alloc_layer(self) -> Dict[str, torch.Tensor]:
        cfg, lay, e, B = self.cfg, self.layout, self.e, self.B
        do, Fo, N, npx, di, Hi, Ho = cfg.embed_dim, cfg.hidden, cfg.seq_len, cfg.n_pixels, cfg.inner_embed_dim, cfg.inner_num_heads, cfg.num_heads
        Mi, Ms, Mo, dap, Fi = self.Mi, self.Ms, self.Mo, lay.dap, lay.Fi
        return {"hi1": e(Mi, di), "qkvi": e(Mi, 3 * dap), "pi": e(Ms * Hi, npx, npx), "oi": e(Mi, dap), "ximid": e(Mi, di), "hi2": e(Mi, di),
                "ui": e(Mi, Fi), "ai": e(Mi, Fi), "outer": e(Mo, do), "ho1": e(Mo, do), "qkvo": e(Mo, 3 * do), "po": e(B * Ho, N, N), "oo": e(Mo, do),
                "xmid": e(Mo, do), "ho2": e(Mo, do), "uo": e(Mo, Fo), "ao": e(Mo, Fo)}

    def init_params(self, seed: int = 0):
        from .tnt_engine import TNTEngine

        TNTEngine.init_params(self, seed)

    def _alloc_saved(self):
        cfg, lay, e = self.cfg, self.layout, self.e
        do, Fo, di, NL = cfg.embed_dim, cfg.hidden, cfg.inner_embed_dim, cfg.num_layers
        Mi, Ms, Mo = self.Mi, self.Ms, self.Mo
        layers = [self._alloc_layer() for _ in range(NL)]
        self.sv = {k: [layers[l][k] for l in range(NL)] for k in self.INNER + self.OUTER}
        self.sv["xi"] = [e(Mi, di) for _ in range(NL + 1)]
        self.sv["xo"] = [e(Mo, do) for _ in range(NL + 1)]
        self.bw = {"dxo": e(Mo, do), "d_outer": e(Mo, do), "d_ao": e(Mo, Fo), "d_ho": e(Mo, do), "d_oo": e(Mo, do), "dqkvo": e(Mo, 3 * do),
                   "d_tok": e(Ms, do), "dxi": e(Mi, di), "d_ai": e(Mi, lay.Fi), "d_hi": e(Mi, di), "d_oi": e(Mi, lay.dap),
                   "dqkvi": e(Mi, 3 * lay.dap)}

    def _build(self, save: bool = False) -> _Plan:
        P, L, cfg, lay = _Plan(), self.L, self.cfg, self.layout
        do, Fo, C, N, NL, n, npx = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers, cfg.n_patches, cfg.n_pixels
        di, Hi, Ho, Fi, dap, HDP, B = cfg.inner_embed_dim, cfg.inner_num_heads, cfg.num_heads, lay.Fi, lay.dap, self.HDP, self.B
        hdi, hdo = di // Hi, do // Ho
        Mi, Ms, Mo = self.Mi, self.Ms, self.Mo
        pp = self._off
        ptr = lambda t: t.data_ptr()  # noqa: E731
        si, so = ptr(self.si), ptr(self.so)
        ln = L.savit_layernorm_fwd_f32
        xi = ptr(self.sv["xi"][0]) if save else ptr(self.xi0)
        xo_t = self.sv["xo"][0] if save else self.xo0
        xo = ptr(xo_t)
        # embeddings: pixel tokens (tnt.py:17-33,155-158) and patch tokens + cls + position (tnt.py:160-167)
        P.add(_gather_pixels, (self,), "pixel_gather")
        self._gemm(P, "pixel_embed", self.pix.data_ptr(), pp("Wpx"), xi, Mi, di, lay.pix_in, lay.pix_in, di, di, bias=pp("bpx"), aux=pp("ppos"), ldaux=di,
                   aux_row_mod=npx)
        P.add(L.savit_patchify_f32, (self.images.data_ptr(), self.patches.data_ptr(), B, cfg.img_size, cfg.patch), "patchify")
        self._gemm(P, "patch_embed", self.patches.data_ptr(), pp("Wpe"), self.tok.data_ptr(), Ms, do, cfg.patch_dim, cfg.patch_dim, do, do, bias=pp("bpa"))
        P.add(L.savit_assemble_tokens_f32, (self.tok.data_ptr(), pp("cls"), pp("pos"), xo, B, N, do), "tokens")
        for l in range(NL):
            p = f"l{l}."
            a = {k: ptr((self.sv[k][l] if save else self.act[k])) for k in self.INNER + self.OUTER}
            # in place without saving; into the next saved buffers with (x_mid = x + ..., x_next = x_mid + ... : the same sums)
            xi_mid, xi_next, xo_next = (a["ximid"], ptr(self.sv["xi"][l + 1]), ptr(self.sv["xo"][l + 1])) if save else (xi, xi, xo)
            res = (lambda src: dict(aux=src, ldaux=di)) if save else (lambda src: dict(accumulate=1))
            # inner block on the pixel stream (tnt.py:68-80): sequences of npx pixel tokens, Hi heads padded to HDP columns
            P.add(ln, (xi, pp(p + "iln1_g"), pp(p + "iln1_b"), a["hi1"], Mi, di, di, di, 1e-6), p + "iln1")
            self._gemm(P, p + "iqkv", a["hi1"], pp(p + "iWqkv"), a["qkvi"], Mi, 3 * dap, di, di, 3 * dap, 3 * dap, alpha=1.0 / math.sqrt(hdi), alpha_cols=dap)
            self._gemm(P, p + "iscores", a["qkvi"], a["qkvi"] + 4 * dap, si, npx, npx, HDP, 3 * dap, 3 * dap, npx, transW=1, batch=Ms * Hi, inner=Hi,
                       sA=(npx * 3 * dap, HDP), sW=(npx * 3 * dap, HDP), sC=(Hi * npx * npx, npx * npx))
            P.add(L.savit_softmax_rows_f32, (si, a["pi"], Ms * Hi * npx, npx, npx), p + "isoftmax")
            self._gemm(P, p + "ipv", a["pi"], a["qkvi"] + 8 * dap, a["oi"], npx, HDP, npx, npx, 3 * dap, dap, batch=Ms * Hi, inner=Hi,
                       sA=(Hi * npx * npx, npx * npx), sW=(npx * 3 * dap, HDP), sC=(npx * dap, HDP))
            self._gemm(P, p + "iproj", a["oi"], pp(p + "iWo"), xi_mid, Mi, di, dap, dap, di, di, **res(xi))
            P.add(ln, (xi_mid, pp(p + "iln2_g"), pp(p + "iln2_b"), a["hi2"], Mi, di, di, di, 1e-6), p + "iln2")
            self._gemm(P, p + "ifc1", a["hi2"], pp(p + "iW1"), a["ai"], Mi, Fi, di, di, Fi, Fi, bias=pp(p + "ib1"), act=1, C2=a["ui"])
            self._gemm(P, p + "ifc2", a["ai"], pp(p + "iW2"), xi_next, Mi, di, Fi, Fi, di, di, bias=pp(p + "ib2"), **res(xi_mid))
            xi = xi_next
            # Inner2Outer (tnt.py:40-51,82-84): outer[b, 1 + j] = pixels[b, j].flatten() Wio + bio + patches[b, 1 + j]; outer[b, 0] = patches[b, 0]
            P.add(_copy_cls_rows, (self, self.sv["outer"][l] if save else self.act["outer"], xo_t), p + "i2o.cls")
            self._gemm(P, p + "i2o", xi, pp(p + "Wio"), a["outer"] + 4 * do, n, do, npx * di, npx * di, do, do, bias=pp(p + "bio"), aux=xo + 4 * do, ldaux=do,
                       batch=B, sA=(n * npx * di, 0), sC=(N * do, 0))
            # outer block (tnt.py:85-92): attention reads the Inner2Outer sum, its residual adds the patch stream (tnt.py:86)
            P.add(ln, (a["outer"], pp(p + "ln1_g"), pp(p + "ln1_b"), a["ho1"], Mo, do, do, do, 1e-6), p + "ln1")
            self._gemm(P, p + "qkv", a["ho1"], pp(p + "Wqkv"), a["qkvo"], Mo, 3 * do, do, do, 3 * do, 3 * do, alpha=1.0 / math.sqrt(hdo), alpha_cols=do)
            self._scores(P, p + "scores", a["qkvo"], a["qkvo"] + 4 * do, self.so, N, N, Ho, hdo, 3 * do, 3 * do, N * 3 * do, N * 3 * do)
            P.add(L.savit_softmax_rows_f32, (so, a["po"], B * Ho * N, N, N), p + "softmax")
            self._gemm(P, p + "pv", a["po"], a["qkvo"] + 8 * do, a["oo"], N, hdo, N, N, 3 * do, do, batch=B * Ho, inner=Ho, sA=(Ho * N * N, N * N),
                       sW=(N * 3 * do, hdo), sC=(N * do, hdo))
            self._gemm(P, p + "proj", a["oo"], pp(p + "Wo"), a["xmid"], Mo, do, do, do, do, do, aux=xo, ldaux=do)
            P.add(ln, (a["xmid"], pp(p + "ln2_g"), pp(p + "ln2_b"), a["ho2"], Mo, do, do, do, 1e-6), p + "ln2")
            self._gemm(P, p + "fc1", a["ho2"], pp(p + "W1"), a["ao"], Mo, Fo, do, do, Fo, Fo, bias=pp(p + "b1"), act=1, C2=a["uo"])
            self._gemm(P, p + "fc2", a["ao"], pp(p + "W2"), xo_next, Mo, do, Fo, Fo, do, do, bias=pp(p + "b2"), aux=a["xmid"], ldaux=do)
            xo = xo_next
            xo_t = self.sv["xo"][l + 1] if save else self.xo0
        self._gemm(P, "head", xo, pp("Wh"), self.logits.data_ptr(), B, C, do, N * do, C, C, bias=pp("bh"))  # cls rows, no final LayerNorm (tnt.py:188-192)
        return P

    def _attention_bwd(self, P: _Plan, label: str, qkv: int, pr: int, d_o: int, dqkv: int, s: int, seqs: int, T: int, H: int, hd: int, width: int,
                       scale: float):
        """dqkv of softmax(q k^T) v for `seqs` sequences of T tokens, H heads of (stored) width hd in rows of 3 * width columns (q already
        carries 1 / sqrt(head_dim): attention.py:41-48): dP = dO V^T ; dV = P^T dO ; dS = P (dP - sum dP P) ; dQ = scale dS K ; dK = dS^T Q."""
        bh = dict(batch=seqs * H, inner=H)
        sS, sQ, sO = (H * T * T, T * T), (T * 3 * width, hd), (T * width, hd)
        self._gemm(P, label + "dP", d_o, qkv + 8 * width, s, T, T, hd, width, 3 * width, T, transW=1, sA=sO, sW=sQ, sC=sS, **bh)
        self._gemm(P, label + "dV", pr, d_o, dqkv + 8 * width, T, hd, T, T, width, 3 * width, transA=1, sA=sS, sW=sO, sC=sQ, **bh)
        P.add(self.L.savit_softmax_rows_bwd_f32, (pr, s, s, seqs * H * T, T, T), label + "softmax.bwd")
        self._gemm(P, label + "dQ", s, qkv + 4 * width, dqkv, T, hd, T, T, 3 * width, 3 * width, sA=sS, sW=sQ, sC=sQ, alpha=scale, alpha_cols=hd, **bh)
        self._gemm(P, label + "dK", s, qkv, dqkv + 4 * width, T, hd, T, T, 3 * width, 3 * width, transA=1, sA=sS, sW=sQ, sC=sQ, **bh)

    def _build_bwd(self) -> _Plan:
        """Reverse-mode gradient of `_build(save=True)` (jax.value_and_grad at train.py:94-95 through tnt.py:150-193): two cotangent
        streams - dxo for the patch stream, dxi for the pixel stream - meet in every layer's Inner2Outer."""
        P, L, cfg, lay, sv, bw = _Plan(), self.L, self.cfg, self.layout, self.sv, self.bw
        do, Fo, C, N, NL, n, npx = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers, cfg.n_patches, cfg.n_pixels
        di, Hi, Ho, Fi, dap, HDP, B = cfg.inner_embed_dim, cfg.inner_num_heads, cfg.num_heads, lay.Fi, lay.dap, self.HDP, self.B
        hdi, hdo = di // Hi, do // Ho
        Mi, Ms, Mo = self.Mi, self.Ms, self.Mo
        pp = self._off
        gp = lambda nm: self._off(nm, self.grads)  # noqa: E731
        ptr = lambda t: t.data_ptr()  # noqa: E731
        dxo, d_outer, d_ao, d_ho, d_oo, dqkvo, d_tok, dxi, d_ai, d_hi, d_oi, dqkvi = (ptr(bw[k]) for k in (
            "dxo", "d_outer", "d_ao", "d_ho", "d_oo", "dqkvo", "d_tok", "dxi", "d_ai", "d_hi", "d_oi", "dqkvi"))
        si, so, dl = ptr(self.si), ptr(self.so), self.dlogits.data_ptr()
        ln_bwd, colsum = L.savit_layernorm_bwd_f32, L.savit_colsum_f32
        # head on the cls rows, no final LayerNorm (tnt.py:188-192); the last layer's pixel stream has no reader but its Inner2Outer
        self._wgrad(P, "head.wgrad", ptr(sv["xo"][NL]), dl, gp("Wh"), B, do, C, N * do, C)
        P.add(colsum, (dl, gp("bh"), B, C, C), "head.bgrad")
        P.add(L.savit_zero_bytes, (dxo, Mo * do * 4), "zero.dxo")
        P.add(L.savit_zero_bytes, (dxi, Mi * di * 4), "zero.dxi")
        self._dgrad(P, "head.dgrad", dl, pp("Wh"), dxo, B, do, C, C, C, N * do)
        for l in range(NL - 1, -1, -1):
            p = f"l{l}."
            a = {k: ptr(sv[k][l]) for k in self.INNER + self.OUTER}
            xi_in, xi_out = ptr(sv["xi"][l]), ptr(sv["xi"][l + 1])
            # outer FFBlock: xo_next = xmid + gelu(ho2 W1 + b1) W2 + b2   (tnt.py:89-92, ff.py:26-33)
            self._wgrad(P, p + "W2.wgrad", a["ao"], dxo, gp(p + "W2"), Mo, Fo, do, Fo, do)
            P.add(colsum, (dxo, gp(p + "b2"), Mo, do, do), p + "b2.grad")
            self._dgrad(P, p + "fc2.dgrad", dxo, pp(p + "W2"), d_ao, Mo, Fo, do, do, do, Fo, act=2, U=a["uo"])
            self._wgrad(P, p + "W1.wgrad", a["ho2"], d_ao, gp(p + "W1"), Mo, do, Fo, do, Fo)
            P.add(colsum, (d_ao, gp(p + "b1"), Mo, Fo, Fo), p + "b1.grad")
            self._dgrad(P, p + "fc1.dgrad", d_ao, pp(p + "W1"), d_ho, Mo, do, Fo, Fo, Fo, do)
            P.add(ln_bwd, (d_ho, a["xmid"], pp(p + "ln2_g"), dxo, dxo, gp(p + "ln2_g"), gp(p + "ln2_b"), Mo, do, do, do, 1e-6), p + "ln2.bwd")
            # outer attention: xmid = xo + attn(LN(outer)) Wo - the residual is the patch stream, the attention reads outer (tnt.py:85-88)
            self._wgrad(P, p + "Wo.wgrad", a["oo"], dxo, gp(p + "Wo"), Mo, do, do, do, do)
            self._dgrad(P, p + "proj.dgrad", dxo, pp(p + "Wo"), d_oo, Mo, do, do, do, do, do)
            self._attention_bwd(P, p, a["qkvo"], a["po"], d_oo, dqkvo, so, B, N, Ho, hdo, do, 1.0 / math.sqrt(hdo))
            self._wgrad(P, p + "Wqkv.wgrad", a["ho1"], dqkvo, gp(p + "Wqkv"), Mo, do, 3 * do, do, 3 * do)
            self._dgrad(P, p + "qkv.dgrad", dqkvo, pp(p + "Wqkv"), d_ho, Mo, do, 3 * do, 3 * do, 3 * do, do)
            P.add(ln_bwd, (d_ho, a["outer"], pp(p + "ln1_g"), None, d_outer, gp(p + "ln1_g"), gp(p + "ln1_b"), Mo, do, do, do, 1e-6), p + "ln1.bwd")
            # Inner2Outer (tnt.py:40-51): outer = patches + pad(flatten(pixels) Wio + bio): every row's cotangent joins the patch stream,
            # the patch rows' also feeds the Dense and, through it, the pixel stream
            P.add(L.savit_add_rows_periodic, (dxo, d_outer, Mo, Mo, do), p + "i2o.dpatches")
            P.add(_gather_patch_rows_tnt, (self, "d_outer"), p + "i2o.gather")
            self._wgrad(P, p + "Wio.wgrad", xi_out, d_tok, gp(p + "Wio"), Ms, npx * di, do, npx * di, do)
            P.add(colsum, (d_tok, gp(p + "bio"), Ms, do, do), p + "bio.grad")
            self._dgrad(P, p + "i2o.dgrad", d_tok, pp(p + "Wio"), dxi, Ms, npx * di, do, do, do, npx * di, accumulate=1)
            # inner FFBlock (tnt.py:76-80)
            self._wgrad(P, p + "iW2.wgrad", a["ai"], dxi, gp(p + "iW2"), Mi, Fi, di, Fi, di)
            P.add(colsum, (dxi, gp(p + "ib2"), Mi, di, di), p + "ib2.grad")
            self._dgrad(P, p + "ifc2.dgrad", dxi, pp(p + "iW2"), d_ai, Mi, Fi, di, di, di, Fi, act=2, U=a["ui"])
            self._wgrad(P, p + "iW1.wgrad", a["hi2"], d_ai, gp(p + "iW1"), Mi, di, Fi, di, Fi)
            P.add(colsum, (d_ai, gp(p + "ib1"), Mi, Fi, Fi), p + "ib1.grad")
            self._dgrad(P, p + "ifc1.dgrad", d_ai, pp(p + "iW1"), d_hi, Mi, di, Fi, Fi, Fi, di)
            P.add(ln_bwd, (d_hi, a["ximid"], pp(p + "iln2_g"), dxi, dxi, gp(p + "iln2_g"), gp(p + "iln2_b"), Mi, di, di, di, 1e-6), p + "iln2.bwd")
            # inner attention over the npx pixel tokens of a patch (tnt.py:68-75), heads of stored width HDP
            self._wgrad(P, p + "iWo.wgrad", a["oi"], dxi, gp(p + "iWo"), Mi, dap, di, dap, di)
            self._dgrad(P, p + "iproj.dgrad", dxi, pp(p + "iWo"), d_oi, Mi, dap, di, di, di, dap)
            self._attention_bwd(P, p + "i", a["qkvi"], a["pi"], d_oi, dqkvi, si, Ms, npx, Hi, HDP, dap, 1.0 / math.sqrt(hdi))
            self._wgrad(P, p + "iWqkv.wgrad", a["hi1"], dqkvi, gp(p + "iWqkv"), Mi, di, 3 * dap, di, 3 * dap)
            self._dgrad(P, p + "iqkv.dgrad", dqkvi, pp(p + "iWqkv"), d_hi, Mi, di, 3 * dap, 3 * dap, 3 * dap, di)
            P.add(ln_bwd, (d_hi, xi_in, pp(p + "iln1_g"), dxi, dxi, gp(p + "iln1_g"), gp(p + "iln1_b"), Mi, di, di, di, 1e-6), p + "iln1.bwd")
        # patch stream: position / cls / patch embedding with bias (tnt.py:160-167)
        P.add(L.savit_pos_cls_grad, (dxo, gp("pos"), gp("cls"), B, N, do, 1), "pos_cls.grad")
        P.add(_gather_patch_rows_tnt, (self, "dxo"), "dtok.gather")
        self._wgrad(P, "Wpe.wgrad", self.patches.data_ptr(), d_tok, gp("Wpe"), Ms, cfg.patch_dim, do, cfg.patch_dim, do)
        P.add(colsum, (d_tok, gp("bpa"), Ms, do, do), "bpa.grad")
        # pixel stream: Dense + bias + the pixel position table shared by every patch (tnt.py:155-158)
        self._wgrad(P, "Wpx.wgrad", self.pix.data_ptr(), dxi, gp("Wpx"), Mi, lay.pix_in, di, lay.pix_in, di)
        P.add(colsum, (dxi, gp("bpx"), Mi, di, di), "bpx.grad")
        P.add(colsum, (dxi, gp("ppos"), Ms, npx * di, npx * di), "ppos.grad")
        return P


def _concat_rows(eng: "CaiTEngineF32", c: int, save: bool, x_final: torch.Tensor, stream: int) -> int:
    """xc = concat([cls, x], axis=1) (cait.py:98): two strided device copies on the current stream (memory plumbing, no arithmetic)."""
    B, N, d = eng.B, eng.cfg.n_patches, eng.cfg.embed_dim
    xc = (eng.sv["xc"][c] if save else eng.xc).view(B, N + 1, d)
    xc[:, 0].copy_(eng.sv["clsin"][c] if save else eng.cls)
    xc[:, 1:].copy_(x_final.view(B, N, d))
    return 0


def _cls_rows_copy(eng: "CaiTEngineF32", compact: str, concat: str, into_concat: bool, stream: int) -> int:
    """Row 0 of every image of a [B (N + 1), d] cotangent <-> the compact [B, d] cls cotangent (strided device copy, no arithmetic)."""
    B, N, d = eng.B, eng.cfg.n_patches, eng.cfg.embed_dim
    rows = eng.bw[concat].view(B, N + 1, d)[:, 0]
    if into_concat:
        rows.copy_(eng.bw[compact])
    else:
        eng.bw[compact].copy_(rows)
    return 0


def _patch_rows_copy(eng: "CaiTEngineF32", stream: int) -> int:
    """dres = rows 1.. of d / d[cls ; x] (the patch rows: the cotangent of the SA stage's output), contiguous."""
    B, N, d = eng.B, eng.cfg.n_patches, eng.cfg.embed_dim
    eng.bw["dres"].view(B, N, d).copy_(eng.bw["dxc"].view(B, N + 1, d)[:, 1:])
    return 0


def _gather_patch_rows(eng: "ViTEngineF32", stream: int) -> int:
    """tok = dx0[:, 1:, :] (the cotangent of the patch embeddings, vit.py:81-84): one strided device copy on the current stream."""
    B, N, d = eng.B, eng.cfg.seq_len, eng.cfg.embed_dim
    eng.tok.view(B, N - 1, d).copy_(eng.dres.view(B, N, d)[:, 1:])
    return 0


def _gather_pixels(eng: "TNTEngineF32", stream: int) -> int:
    """PixelEmbedBlock's rearrangement (tnt.py:24-29): every patch as its (P / t)^2 pixel tokens of c t t features - one permuted device
    copy on the current stream (memory plumbing, no arithmetic)."""
    cfg, B = eng.cfg, eng.B
    g, t = cfg.img_size // cfg.patch, cfg.transformed_patch
    s = cfg.patch // t
    eng.pix.view(B, g, g, s, s, 3, t, t).copy_(eng.images.view(B, g, s, t, g, s, t, 3).permute(0, 1, 4, 2, 5, 7, 3, 6))
    return 0


def _copy_cls_rows(eng: "TNTEngineF32", outer: torch.Tensor, xo: torch.Tensor, stream: int) -> int:
    """outer[:, 0] = patches[:, 0]: Inner2Outer pads a zero row for cls (tnt.py:48-50).  One strided device copy (memory plumbing)."""
    B, N, do = eng.B, eng.cfg.seq_len, eng.cfg.embed_dim
    outer.view(B, N, do)[:, 0].copy_(xo.view(B, N, do)[:, 0])
    return 0


def _gather_patch_rows_tnt(eng: "TNTEngineF32", src: str, stream: int) -> int:
    """d_tok = cotangent[:, 1:, :] (the patch rows of every image, contiguous): one strided device copy on the current stream."""
    B, N, do = eng.B, eng.cfg.seq_len, eng.cfg.embed_dim
    eng.bw["d_tok"].view(B, N - 1, do).copy_(eng.bw[src].view(B, N, do)[:, 1:])
    return 0
