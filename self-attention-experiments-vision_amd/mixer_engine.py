"""MLP-Mixer training engine: the ViT engine's launch-plan machinery, GEMM / LayerNorm / loss / optimizer
kernels and HBM conventions around `MLPMixer.__call__` (/root/reference/models/mlp_mixer.py:44-64, MixerBlock :17-31).

What differs from the ViT path
  * token mixing: FFBlock over the TOKEN axis (mlp_mixer.py:19-23).  The LayerNorm output [B, l, d] is transposed per image
    into [B*d, Lp] (savit_transpose_bf16), the two Dense layers are the same TN GEMMs with M = B*d rows, and the way back
    is the transpose fused with `x = x + inputs` (:24).  l = 196 / 49 tokens and int(0.5 l) = 98 / 24 hidden units are not
    multiples of the MFMA K-step, so BOTH are padded to multiples of 64 with zeros - in the activations (pad columns are never
    written with anything but zeros) AND in the parameter buffer: the token kernels are STORED [Lp, Fp] / [Fp, Lp] and biases
    [Fp] / [Lp]; the Flax tree exposes the logical [l, F] / [F, l] / [F] / [l] corners as strided views.  A pad weight only
    ever meets a zero activation or a zero cotangent, so its gradient is exactly 0 and AdamW (decay * 0, 0 / (0 + eps))
    keeps it 0: the padding is invisible to the arithmetic.
  * no cls token / position embedding; patch embedding has a bias (:46-49); the stream between blocks stays in the module
    dtype (bf16 sums, :24,:30) - kept as bf16-representable values in the fp32 residual buffers the shared kernels use;
  * head: LayerNorm over every token, mean over tokens (:61-62), Dense with the default (non-zero) initialiser (:63).
"""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import lib as _lib
from .config import ModelConfig
from .engine import ViTEngine, _Plan, _align, _copy_tree, bf16, f32, finalize_wgrad_ws  # noqa: F401
from .options import EngineOptions


class MixerLayout:
    """Offsets (fp32 elements) in the flat parameter buffer; `off` holds STORAGE shapes, `logical` the reference's shapes."""

    def __init__(self, cfg: ModelConfig):
        if cfg.kind != "mixer":
            raise NotImplementedError("MixerLayout lays out the MLP-Mixer family")
        self.cfg = cfg
        d, F, C, n, L = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.n_patches, cfg.num_layers
        Ft = cfg.tokens_hidden
        self.Lp, self.Fp = _align(n, 64), _align(Ft, 64)
        Lp, Fp = self.Lp, self.Fp
        self.off: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.logical: Dict[str, Tuple[int, ...]] = {}
        cur = 0

        def add(name, shape, logical=None):
            nonlocal cur
            k = 1
            for s in shape:
                k *= s
            self.off[name] = (cur, tuple(shape))
            self.logical[name] = tuple(logical or shape)
            cur += _align(k, 4)

        add("Wpe", (cfg.patch_dim, d))
        add("bpe", (d,))
        cur = _align(cur, 64)
        self.embed_end = cur
        self.layer_start: List[int] = []
        for l in range(L):
            self.layer_start.append(cur)
            add(f"l{l}.ln1_g", (d,))
            add(f"l{l}.ln1_b", (d,))
            add(f"l{l}.tW1", (Lp, Fp), (n, Ft))
            add(f"l{l}.tb1", (Fp,), (Ft,))
            add(f"l{l}.tW2", (Fp, Lp), (Ft, n))
            add(f"l{l}.tb2", (Lp,), (n,))
            add(f"l{l}.ln2_g", (d,))
            add(f"l{l}.ln2_b", (d,))
            add(f"l{l}.W1", (d, F))
            add(f"l{l}.b1", (F,))
            add(f"l{l}.W2", (F, d))
            add(f"l{l}.b2", (d,))
            cur = _align(cur, 64)
        self.layer_stride = (self.layer_start[1] - self.layer_start[0]) if L > 1 else (cur - self.layer_start[0])
        self.final_start = cur
        add("lnf_g", (d,))
        add("lnf_b", (d,))
        add("Wh", (d, C))
        add("bh", (C,))
        self.total = _align(cur, 64)

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        """The reference-shaped tensor: a (strided, for the padded token-mixing parameters) view into `flat`."""
        o, shape = self.off[name]
        k = 1
        for s in shape:
            k *= s
        t = flat[o:o + k].view(*shape)
        for ax, s in enumerate(self.logical[name]):
            t = t.narrow(ax, 0, s)
        return t

    def flax_tree(self, flat: torch.Tensor) -> dict:
        """Flax-shaped nested dict of VIEWS (module names as flax.linen auto-numbers them inside MLPMixer / MixerBlock)."""
        v = lambda n: self.view(flat, n)  # noqa: E731
        p = {"PatchEmbedBlock_0": {"Dense_0": {"kernel": v("Wpe"), "bias": v("bpe")}}}
        for l in range(self.cfg.num_layers):
            p[f"MixerBlock_{l}"] = {
                "LayerNorm_0": {"scale": v(f"l{l}.ln1_g"), "bias": v(f"l{l}.ln1_b")},
                "FFBlock_0": {"Dense_0": {"kernel": v(f"l{l}.tW1"), "bias": v(f"l{l}.tb1")},
                              "Dense_1": {"kernel": v(f"l{l}.tW2"), "bias": v(f"l{l}.tb2")}},
                "LayerNorm_1": {"scale": v(f"l{l}.ln2_g"), "bias": v(f"l{l}.ln2_b")},
                "FFBlock_1": {"Dense_0": {"kernel": v(f"l{l}.W1"), "bias": v(f"l{l}.b1")},
                              "Dense_1": {"kernel": v(f"l{l}.W2"), "bias": v(f"l{l}.b2")}},
            }
        p["LayerNorm_0"] = {"scale": v("lnf_g"), "bias": v("lnf_b")}
        p["Dense_0"] = {"kernel": v("Wh"), "bias": v("bh")}
        return {"params": p}


class MixerEngine(ViTEngine):
    """Same public surface as ViTEngine (forward / loss_backward / optimizer_step / profile_step / bwd_hooks)."""

    DEFAULT_OVERLAP = True  # many small launches: the side stream still pays (engine.ViTEngine._init_step_state)

    def __init__(self, cfg: ModelConfig, batch: int, device: str = "cuda", round_like_reference: bool = True,
                 reserved_cus=None, wgrad_max_lag=None, options=None, **opts):
        self.opt = EngineOptions.resolve(options, reserved_cus=reserved_cus, wgrad_max_lag=wgrad_max_lag, **opts)
        if cfg.kind != "mixer":
            raise NotImplementedError("MixerEngine handles the MLP-Mixer family")
        if cfg.embed_dim % 64 != 0 or cfg.patch % 8 != 0 or cfg.num_classes % 8 != 0:
            raise ValueError("embed_dim % 64, patch % 8 and num_classes % 8 must be 0")
        if not torch.cuda.is_available():
            raise RuntimeError("MixerEngine needs a GPU: there is no CPU path")
        self.L = _lib.load()
        self.cfg = cfg
        self.B = int(batch)
        self.dev = torch.device(device)
        self.rp = int(round_like_reference)
        self._init_cu_budget()
        self.layout = MixerLayout(cfg)
        d, F, C, n, NL = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.n_patches, cfg.num_layers
        Lp, Fp = self.layout.Lp, self.layout.Fp
        self.M = M = self.B * n      # token rows
        self.Md = Md = self.B * d    # channel rows of the transposed (token-mixing) activations
        self.Cp = _align(C, 64)
        z = lambda *s, dt=f32: torch.zeros(*s, dtype=dt, device=self.dev)  # noqa: E731
        e = lambda *s, dt=f32: torch.empty(*s, dtype=dt, device=self.dev)  # noqa: E731
        self._init_flat_buffers()
        self.w = {
            "tW1_n": e(NL, Lp, Fp, dt=bf16), "tW1_t": e(NL, Fp, Lp, dt=bf16),
            "tW2_n": e(NL, Fp, Lp, dt=bf16), "tW2_t": e(NL, Lp, Fp, dt=bf16),
            "W1_n": e(NL, d, F, dt=bf16), "W1_t": e(NL, F, d, dt=bf16),
            "W2_n": e(NL, F, d, dt=bf16), "W2_t": e(NL, d, F, dt=bf16),
            "Wpe_t": e(d, cfg.patch_dim, dt=bf16),
            "Wh_t": e(C, d, dt=bf16), "Wh_n": z(d, self.Cp, dt=bf16),
        }
        # ---- activations saved for backward.  Token-mixing tensors are [B*d, Lp] / [B*d, Fp]; their pad columns are zero:
        # allocated zeroed and never written (h1T), or written by a GEMM whose pad weights and biases are zero.
        self.x = [e(M, d) for _ in range(NL + 1)]
        self.xmid = [e(M, d) for _ in range(NL)]
        self.h1T = [z(Md, Lp, dt=bf16) for _ in range(NL)]
        self.tu = [e(Md, Fp, dt=bf16) for _ in range(NL)]
        self.ta = [e(Md, Fp, dt=bf16) for _ in range(NL)]
        self.h2 = [e(M, d, dt=bf16) for _ in range(NL)]
        self.u = [e(M, F, dt=bf16) for _ in range(NL)]
        self.a = [e(M, F, dt=bf16) for _ in range(NL)]
        self.stats = [e(4, M) for _ in range(NL)]
        self.h1 = e(M, d, dt=bf16)        # LayerNorm output before the transpose (scratch; also the final LayerNorm's output)
        self.yT = e(Md, Lp, dt=bf16)      # token-mixing branch before the transpose back
        self.zcls = e(self.B, d, dt=bf16)  # pooled features (name shared with the ViT engine)
        self.fstats = e(2, M)
        # ---- backward scratch (rotated where a side-stream weight-gradient GEMM reads it, as in the ViT engine)
        self.dres = e(M, d)
        depth = max(2, self.opt.ring_depth)
        self.dres_b_ring = [e(M, d, dt=bf16) for _ in range(2 * depth)]
        self.dres_b = self.dres_b_ring[0]
        self.d_u_ring = [e(M, F, dt=bf16) for _ in range(depth)]
        self.dyT_ring = [z(Md, Lp, dt=bf16) for _ in range(depth)]   # pad columns stay zero: the transpose writes l of Lp columns
        self.d_tu_ring = [e(Md, Fp, dt=bf16) for _ in range(depth)]
        self.dhT = e(Md, Lp, dt=bf16)
        self.d_h = e(M, d, dt=bf16)
        self.colsum_slab = e(max(1, self.L.savit_gemm_colsum_rows_cus(M, F, d, 0, self.cu_budget if self.reserved_cus else 0)), F)
        self.tcolsum_slab = e(max(1, self.L.savit_gemm_colsum_rows_cus(Md, Fp, Lp, 0, self.cu_budget if self.reserved_cus else 0)), Fp)
        self.trowsum_slab = z(max(1, self.L.savit_transpose_rowsum_rows(self.B, d)), Lp)  # pad columns stay zero
        self.d_z = e(self.B, d, dt=bf16)
        ws = self.L.savit_layernorm_bwd_workspace_bytes(M, d)
        self.ln_ws = torch.empty(max(int(ws), 16), dtype=torch.uint8, device=self.dev)
        self._init_step_state()

    # ------------------------------------------------------------------------------------ parameters
    def init_params(self, seed: int = 0):
        """flax defaults everywhere (mlp_mixer.py never overrides an initialiser): lecun-normal Dense kernels, zero biases,
        LayerNorm scale 1 / bias 0.  The padding of the token-mixing parameters stays zero."""
        g = torch.Generator(device="cpu").manual_seed(int(seed))
        self.params.zero_()
        lay, cfg = self.layout, self.cfg

        def lecun(name):
            shape = lay.logical[name]
            std = math.sqrt(1.0 / shape[0]) / 0.87962566103423978
            t = torch.empty(shape, dtype=f32)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=g)
            lay.view(self.params, name).copy_(t)

        lecun("Wpe")
        for l in range(cfg.num_layers):
            lay.view(self.params, f"l{l}.ln1_g").fill_(1.0)
            lay.view(self.params, f"l{l}.ln2_g").fill_(1.0)
            for nme in ("tW1", "tW2", "W1", "W2"):
                lecun(f"l{l}.{nme}")
        lay.view(self.params, "lnf_g").fill_(1.0)
        lecun("Wh")
        self.weights_stale = True

    def load_params(self, tree: dict):
        src = tree["params"] if "params" in tree else tree
        _copy_tree(self.param_tree()["params"], src)  # logical views: the padding is not touched and stays zero
        self.weights_stale = True

    # ------------------------------------------------------------------------------------ plans
    def _build_cast_plan(self) -> _Plan:
        P, L, lay, cfg = _Plan(), self.L, self.layout, self.cfg
        d, F, C, NL = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.num_layers
        Lp, Fp = lay.Lp, lay.Fp
        ls = lay.layer_stride
        for name, R, Cc in (("tW1", Lp, Fp), ("tW2", Fp, Lp), ("W1", d, F), ("W2", F, d)):
            P.add(L.savit_cast_transpose_bf16, (self._off_ptr(self.params, f"l0.{name}"), ls, NL, R, Cc, self.w[name + "_n"].data_ptr(),
                                                R * Cc, Cc, self.w[name + "_t"].data_ptr(), R * Cc, R), f"cast {name}")
        P.add(L.savit_cast_transpose_bf16, (self._off_ptr(self.params, "Wpe"), 0, 1, cfg.patch_dim, d, None, 0, d,
                                            self.w["Wpe_t"].data_ptr(), 0, cfg.patch_dim), "cast Wpe")
        P.add(L.savit_cast_transpose_bf16, (self._off_ptr(self.params, "Wh"), 0, 1, d, C, self.w["Wh_n"].data_ptr(), 0, self.Cp,
                                            self.w["Wh_t"].data_ptr(), 0, d), "cast Wh")
        return P

    def _build_fwd_plan(self) -> _Plan:
        P, L, cfg, lay = _Plan(), self.L, self.cfg, self.layout
        d, F, C, n, NL, B, M, Md = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.n_patches, cfg.num_layers, self.B, self.M, self.Md
        Lp, Fp = lay.Lp, lay.Fp
        pp = lambda nme: self._off_ptr(self.params, nme)  # noqa: E731
        x = self.x
        # patch embedding with bias, no position embedding (mlp_mixer.py:46-49, patch_embed.py:19-25)
        self._gemm(P, "patch_embed", A=self._img_buf.data_ptr(), Bt=self.w["Wpe_t"].data_ptr(), C=x[0].data_ptr(), bias=pp("bpe"),
                   M=M, N=d, K=cfg.patch_dim, lda=0, ldb=cfg.patch_dim, ldc=d, epilogue=_lib.EPI_PATCH, img_size=cfg.img_size,
                   patch=cfg.patch, tokens=n, token_offset=0)
        for l in range(NL):
            st = self.stats[l]
            w = lambda nme, l=l: self.w[nme][l].data_ptr()  # noqa: E731
            # token mixing (mlp_mixer.py:18-24)
            P.add(L.savit_layernorm_fwd, (x[l].data_ptr(), pp(f"l{l}.ln1_g"), pp(f"l{l}.ln1_b"), self.h1.data_ptr(), st[0].data_ptr(),
                                          st[1].data_ptr(), M, d, d, 1e-6, self.rp), f"l{l}.ln1")
            P.add(L.savit_transpose_bf16, (self.h1.data_ptr(), n * d, d, self.h1T[l].data_ptr(), d * Lp, Lp, B, n, d, None, None, 0, None, 0),
                  f"l{l}.tok.T")
            self._gemm(P, f"l{l}.tok.fc1", A=self.h1T[l].data_ptr(), Bt=w("tW1_t"), C=self.tu[l].data_ptr(), C2=self.ta[l].data_ptr(),
                       bias=pp(f"l{l}.tb1"), M=Md, N=Fp, K=Lp, lda=Lp, ldb=Lp, ldc=Fp, epilogue=_lib.EPI_BIAS_GELU)
            self._gemm(P, f"l{l}.tok.fc2", A=self.ta[l].data_ptr(), Bt=w("tW2_t"), C=self.yT.data_ptr(), bias=pp(f"l{l}.tb2"), M=Md, N=Lp,
                       K=Fp, lda=Fp, ldb=Fp, ldc=Lp, epilogue=_lib.EPI_BF16)
            P.add(L.savit_transpose_bf16, (self.yT.data_ptr(), d * Lp, Lp, None, 0, d, B, d, n, x[l].data_ptr(), self.xmid[l].data_ptr(),
                                           self.rp, None, 0), f"l{l}.tok.T+res")
            # channel mixing (mlp_mixer.py:26-30)
            P.add(L.savit_layernorm_fwd, (self.xmid[l].data_ptr(), pp(f"l{l}.ln2_g"), pp(f"l{l}.ln2_b"), self.h2[l].data_ptr(),
                                          st[2].data_ptr(), st[3].data_ptr(), M, d, d, 1e-6, self.rp), f"l{l}.ln2")
            self._gemm(P, f"l{l}.fc1", A=self.h2[l].data_ptr(), Bt=w("W1_t"), C=self.u[l].data_ptr(), C2=self.a[l].data_ptr(),
                       bias=pp(f"l{l}.b1"), M=M, N=F, K=d, lda=d, ldb=d, ldc=F, epilogue=_lib.EPI_BIAS_GELU)
            self._gemm(P, f"l{l}.fc2", A=self.a[l].data_ptr(), Bt=w("W2_t"), C=x[l + 1].data_ptr(), bias=pp(f"l{l}.b2"),
                       aux=self.xmid[l].data_ptr(), M=M, N=d, K=F, lda=F, ldb=F, ldc=d, ldaux=d, epilogue=_lib.EPI_RESID,
                       round_out_bf16=self.rp)
        # LayerNorm over every token, mean over tokens, head (mlp_mixer.py:61-63)
        P.add(L.savit_layernorm_fwd, (x[NL].data_ptr(), pp("lnf_g"), pp("lnf_b"), self.h1.data_ptr(), self.fstats[0].data_ptr(),
                                      self.fstats[1].data_ptr(), M, d, d, 1e-6, self.rp), "lnf")
        P.add(L.savit_token_mean_fwd, (self.h1.data_ptr(), self.zcls.data_ptr(), B, n, d), "pool")
        self._gemm(P, "head", A=self.zcls.data_ptr(), Bt=self.w["Wh_t"].data_ptr(), C=self.logits.data_ptr(), bias=pp("bh"), M=B, N=C, K=d,
                   lda=d, ldb=d, ldc=C, epilogue=_lib.EPI_F32, round_out_bf16=self.rp)
        return P

    def _record_bwd_plan(self) -> _Plan:
        P, L, cfg, lay = _Plan(), self.L, self.cfg, self.layout
        d, F, C, n, NL, B, M, Md = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.n_patches, cfg.num_layers, self.B, self.M, self.Md
        Lp, Fp = lay.Lp, lay.Fp
        pp = lambda nme: self._off_ptr(self.params, nme)  # noqa: E731
        gp = lambda nme: self._off_ptr(self.grads, nme)  # noqa: E731
        ws, wsb = self.ln_ws.data_ptr(), self.ln_ws.numel()

        def wgrad(label, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, patch=(0, 0, 0, 0)):
            self._add_wgrad(P, label, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, self._wgrad_splits(Kin, Nout, patch[0]), patch)

        ring, ri = [t.data_ptr() for t in self.dres_b_ring], 0
        # ---- head, mean over tokens, final LayerNorm (every row receives dz / l)
        wgrad("head.wgrad", self.zcls.data_ptr(), self.dlogits.data_ptr(), gp("Wh"), B, d, C, d, self.Cp, C)
        self._gemm(P, "head.dgrad", A=self.dlogits.data_ptr(), Bt=self.w["Wh_n"].data_ptr(), C=self.d_z.data_ptr(), M=B, N=d, K=self.Cp,
                   lda=self.Cp, ldb=self.Cp, ldc=d, epilogue=_lib.EPI_BF16)
        P.add(L.savit_token_mean_bwd, (self.d_z.data_ptr(), self.d_h.data_ptr(), B, n, d), "pool.bwd")
        P.add(L.savit_layernorm_bwd, (self.d_h.data_ptr(), self.x[NL].data_ptr(), pp("lnf_g"), self.fstats[0].data_ptr(),
                                      self.fstats[1].data_ptr(), None, self.dres.data_ptr(), ring[0], gp("lnf_g"), gp("lnf_b"),
                                      gp(f"l{NL - 1}.b2"), M, d, d, d, self.rp, ws, wsb), "lnf.bwd", writes=(ring[0],))
        for l in range(NL - 1, -1, -1):
            st = self.stats[l]
            w = lambda nme, l=l: self.w[nme][l].data_ptr()  # noqa: E731
            k = l % len(self.d_u_ring)
            d_u, dyT, d_tu = self.d_u_ring[k].data_ptr(), self.dyT_ring[k].data_ptr(), self.d_tu_ring[k].data_ptr()
            # channel mixing backward (ff.py:26-33 on [B*l, d])
            wgrad(f"l{l}.W2.wgrad", self.a[l].data_ptr(), ring[ri], gp(f"l{l}.W2"), M, F, d, F, d, d)
            self._gemm(P, f"l{l}.fc2.dgrad", writes=(d_u,), A=ring[ri], Bt=w("W2_n"), C=d_u, aux=self.u[l].data_ptr(),
                       colsum=self.colsum_slab.data_ptr(), colsum_rows=self.colsum_slab.shape[0], M=M, N=F, K=d, lda=d, ldb=d, ldc=F,
                       ldaux=F, epilogue=_lib.EPI_DGELU)
            P.add(L.savit_colsum_finalize, (self.colsum_slab.data_ptr(), self.colsum_slab.shape[0], F, gp(f"l{l}.b1"), 1), f"l{l}.b1.grad")
            wgrad(f"l{l}.W1.wgrad", self.h2[l].data_ptr(), d_u, gp(f"l{l}.W1"), M, d, F, d, F, F)
            self._gemm(P, f"l{l}.fc1.dgrad", A=d_u, Bt=w("W1_n"), C=self.d_h.data_ptr(), M=M, N=d, K=F, lda=F, ldb=F, ldc=d,
                       epilogue=_lib.EPI_BF16)
            ri = (ri + 1) % len(ring)
            P.add(L.savit_layernorm_bwd, (self.d_h.data_ptr(), self.xmid[l].data_ptr(), pp(f"l{l}.ln2_g"), st[2].data_ptr(), st[3].data_ptr(),
                                          self.dres.data_ptr(), self.dres.data_ptr(), ring[ri], gp(f"l{l}.ln2_g"), gp(f"l{l}.ln2_b"), None,
                                          M, d, d, d, self.rp, ws, wsb), f"l{l}.ln2.bwd", writes=(ring[ri],))
            # token mixing backward: cotangent of x_mid transposed per image; its row sums are the gradient of the second
            # token Dense's bias (one value per token, summed over images and channels)
            rs = self.trowsum_slab
            P.add(L.savit_transpose_bf16, (ring[ri], n * d, d, dyT, d * Lp, Lp, B, n, d, None, None, 0, rs.data_ptr(), Lp), f"l{l}.tok.dT",
                  writes=(dyT,))
            P.add(L.savit_colsum_finalize, (rs.data_ptr(), rs.shape[0], Lp, gp(f"l{l}.tb2"), 1), f"l{l}.tb2.grad")
            wgrad(f"l{l}.tW2.wgrad", self.ta[l].data_ptr(), dyT, gp(f"l{l}.tW2"), Md, Fp, Lp, Fp, Lp, Lp)
            self._gemm(P, f"l{l}.tok.fc2.dgrad", writes=(d_tu,), A=dyT, Bt=w("tW2_n"), C=d_tu, aux=self.tu[l].data_ptr(),
                       colsum=self.tcolsum_slab.data_ptr(), colsum_rows=self.tcolsum_slab.shape[0], M=Md, N=Fp, K=Lp, lda=Lp, ldb=Lp,
                       ldc=Fp, ldaux=Fp, epilogue=_lib.EPI_DGELU)
            P.add(L.savit_colsum_finalize, (self.tcolsum_slab.data_ptr(), self.tcolsum_slab.shape[0], Fp, gp(f"l{l}.tb1"), 1),
                  f"l{l}.tb1.grad")
            wgrad(f"l{l}.tW1.wgrad", self.h1T[l].data_ptr(), d_tu, gp(f"l{l}.tW1"), Md, Lp, Fp, Lp, Fp, Fp)
            self._gemm(P, f"l{l}.tok.fc1.dgrad", A=d_tu, Bt=w("tW1_n"), C=self.dhT.data_ptr(), M=Md, N=Lp, K=Fp, lda=Fp, ldb=Fp, ldc=Lp,
                       epilogue=_lib.EPI_BF16)
            P.add(L.savit_transpose_bf16, (self.dhT.data_ptr(), d * Lp, Lp, self.d_h.data_ptr(), n * d, d, B, d, n, None, None, 0, None, 0),
                  f"l{l}.tok.dT.back")
            ri = (ri + 1) % len(ring)
            P.add(L.savit_layernorm_bwd, (self.d_h.data_ptr(), self.x[l].data_ptr(), pp(f"l{l}.ln1_g"), st[0].data_ptr(), st[1].data_ptr(),
                                          self.dres.data_ptr(), self.dres.data_ptr(), ring[ri], gp(f"l{l}.ln1_g"), gp(f"l{l}.ln1_b"),
                                          gp(f"l{l - 1}.b2") if l > 0 else gp("bpe"), M, d, d, d, self.rp, ws, wsb), f"l{l}.ln1.bwd",
                  writes=(ring[ri],))
        wgrad("Wpe.wgrad", self._img_buf.data_ptr(), ring[ri], gp("Wpe"), M, cfg.patch_dim, d, 0, d, d, patch=(cfg.patch, cfg.img_size, n, 0))
        finalize_wgrad_ws(self, P)
        return P

    def activation_bytes(self) -> int:
        tot = 0
        for group in (self.x, self.xmid, self.h1T, self.tu, self.ta, self.h2, self.u, self.a, self.stats):
            tot += sum(t.numel() * t.element_size() for t in group)
        return tot
