"""TNT (Transformer-in-Transformer) training engine: `TNT.__call__` of /root/reference/models/tnt.py:150-193 with
the ViT engine's plan machinery and kernels.  Two residual streams, both fp32 (each receives an fp32 position embedding):
  pixels  [B*n*16, di]   every 16x16 patch is a sequence of 16 "pixel tokens" of width di = 24 / 40  (inner transformer)
  patches [B*(n+1), do]  the usual cls + patch tokens of width do = 384 / 640                          (outer transformer)
Per layer (EncoderBlock, tnt.py:66-93): a pre-LN block on the pixel stream, Inner2Outer (flatten a patch's 16 pixel tokens, Dense
to do, zero row for cls, add to the patch stream), then a pre-LN block on the patch stream whose attention reads the Inner2Outer
sum while its residual adds `patch_inputs` (tnt.py:86).  No final LayerNorm; zero-initialised head on the cls row.

The inner transformer runs through the SAME kernels as the outer one:
  * widths 24 / 40 are not multiples of the 32-deep MFMA K-step: the TN GEMM reads such an operand with K rounded up and
    lda = true width (the extra columns alias the next row) against weight copies that hold ZEROS in those K columns;
  * inner heads are 6 / 10 wide: the q/k/v kernels are STORED head-padded to 16 columns ([di, 3*4*16], zeros in the pad; out
    kernel [4*16, di], zero pad rows) and the fused attention kernels run with head_dim 16 on B*n "images" of 16 tokens.  A pad
    weight only meets zero activations / cotangents, so its gradient is exactly 0 and AdamW keeps it 0 (as for MLP-Mixer);
    the Flax tree exposes the logical [di, 4, hd] / [4, hd, di] corners as strided views.
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

HDP = 16  # padded inner head width


class TNTLayout:
    """Offsets (fp32 elements) in the flat parameter buffer.  `off` holds storage shapes; q/k/v/out kernels of the inner attention
    are stored head-padded (see module docstring)."""

    def __init__(self, cfg: ModelConfig):
        if cfg.kind != "tnt":
            raise NotImplementedError("TNTLayout lays out the TNT family")
        self.cfg = cfg
        do, Fo, C, N, L = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers
        di, Hi, npx = cfg.inner_embed_dim, cfg.inner_num_heads, cfg.n_pixels
        self.dap = Hi * HDP                       # padded inner attention width
        self.Fi = max(1, int(4 * di))             # inner FFBlock hidden (ff.py:24, expand 4)
        self.pix_in = 3 * cfg.transformed_patch ** 2
        self.off: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        cur = 0

        def add(name, shape):
            nonlocal cur
            k = 1
            for s in shape:
                k *= s
            self.off[name] = (cur, tuple(shape))
            cur += _align(k, 4)

        add("Wpe", (cfg.patch_dim, do))
        add("bpa", (do,))
        add("cls", (do,))
        add("pos", (N, do))
        add("Wpx", (self.pix_in, di))
        add("bpx", (di,))
        add("ppos", (npx, di))
        cur = _align(cur, 64)
        self.embed_end = cur
        self.layer_start: List[int] = []
        for l in range(L):
            self.layer_start.append(cur)
            p = f"l{l}."
            add(p + "iln1_g", (di,)); add(p + "iln1_b", (di,))
            add(p + "iWqkv", (di, 3 * self.dap))
            add(p + "iWo", (self.dap, di))
            add(p + "iln2_g", (di,)); add(p + "iln2_b", (di,))
            add(p + "iW1", (di, self.Fi)); add(p + "ib1", (self.Fi,))
            add(p + "iW2", (self.Fi, di)); add(p + "ib2", (di,))
            add(p + "Wio", (npx * di, do)); add(p + "bio", (do,))
            add(p + "ln1_g", (do,)); add(p + "ln1_b", (do,))
            add(p + "Wqkv", (do, 3 * do))
            add(p + "Wo", (do, do))
            add(p + "ln2_g", (do,)); add(p + "ln2_b", (do,))
            add(p + "W1", (do, Fo)); add(p + "b1", (Fo,))
            add(p + "W2", (Fo, do)); add(p + "b2", (do,))
            cur = _align(cur, 64)
        self.layer_stride = (self.layer_start[1] - self.layer_start[0]) if L > 1 else (cur - self.layer_start[0])
        self.final_start = cur
        add("Wh", (do, C))
        add("bh", (C,))
        self.total = _align(cur, 64)

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        o, shape = self.off[name]
        k = 1
        for s in shape:
            k *= s
        return flat[o:o + k].view(*shape)

    def flax_tree(self, flat: torch.Tensor) -> dict:
        cfg = self.cfg
        do, Ho, di, Hi = cfg.embed_dim, cfg.num_heads, cfg.inner_embed_dim, cfg.inner_num_heads
        hdo, hdi = do // Ho, di // Hi
        v = lambda n: self.view(flat, n)  # noqa: E731
        enc = {}
        for l in range(cfg.num_layers):
            p = f"l{l}."
            iw = v(p + "iWqkv").view(di, 3, Hi, HDP)[:, :, :, :hdi]     # logical corner of the head-padded storage
            ow = v(p + "Wqkv")
            enc[f"EncoderBlock_{l}"] = {
                "LayerNorm_0": {"scale": v(p + "iln1_g"), "bias": v(p + "iln1_b")},
                "SelfAttentionBlock_0": {
                    "queries": {"kernel": iw[:, 0]}, "keys": {"kernel": iw[:, 1]}, "values": {"kernel": iw[:, 2]},
                    "DenseGeneral_0": {"kernel": v(p + "iWo").view(Hi, HDP, di)[:, :hdi, :]}},
                "LayerNorm_1": {"scale": v(p + "iln2_g"), "bias": v(p + "iln2_b")},
                "FFBlock_0": {"Dense_0": {"kernel": v(p + "iW1"), "bias": v(p + "ib1")},
                              "Dense_1": {"kernel": v(p + "iW2"), "bias": v(p + "ib2")}},
                "Inner2OuterBlock_0": {"Dense_0": {"kernel": v(p + "Wio"), "bias": v(p + "bio")}},
                "LayerNorm_2": {"scale": v(p + "ln1_g"), "bias": v(p + "ln1_b")},
                "SelfAttentionBlock_1": {
                    "queries": {"kernel": ow[:, 0:do].unflatten(1, (Ho, hdo))},
                    "keys": {"kernel": ow[:, do:2 * do].unflatten(1, (Ho, hdo))},
                    "values": {"kernel": ow[:, 2 * do:3 * do].unflatten(1, (Ho, hdo))},
                    "DenseGeneral_0": {"kernel": v(p + "Wo").view(Ho, hdo, do)}},
                "LayerNorm_3": {"scale": v(p + "ln2_g"), "bias": v(p + "ln2_b")},
                "FFBlock_1": {"Dense_0": {"kernel": v(p + "W1"), "bias": v(p + "b1")},
                              "Dense_1": {"kernel": v(p + "W2"), "bias": v(p + "b2")}},
            }
        return {"params": {
            "PixelEmbedBlock_0": {"Dense_0": {"kernel": v("Wpx"), "bias": v("bpx")}},
            "PatchEmbedBlock_0": {"Dense_0": {"kernel": v("Wpe"), "bias": v("bpa")}},
            "cls": v("cls").view(1, 1, do),
            "AddAbsPosEmbed_0": {"pos_embed": v("ppos").view(1, cfg.n_pixels, di)},
            "AddAbsPosEmbed_1": {"pos_embed": v("pos").view(1, cfg.seq_len, do)},
            "Encoder_0": enc,
            "Dense_0": {"kernel": v("Wh"), "bias": v("bh")},
        }}


class TNTEngine(ViTEngine):
    """Same public surface as ViTEngine (forward / loss_backward / optimizer_step / profile_step / bwd_hooks)."""

    DEFAULT_OVERLAP = True  # many small launches: the side stream still pays (engine.ViTEngine._init_step_state)

    def __init__(self, cfg: ModelConfig, batch: int, device: str = "cuda", round_like_reference: bool = True,
                 reserved_cus=None, wgrad_max_lag=None, options=None, **opts):
        self.opt = EngineOptions.resolve(options, reserved_cus=reserved_cus, wgrad_max_lag=wgrad_max_lag, **opts)
        if cfg.kind != "tnt":
            raise NotImplementedError("TNTEngine handles the TNT family")
        if cfg.head_dim != 64 or cfg.embed_dim % 64 or cfg.num_classes % 8 or cfg.patch % 8:
            raise ValueError("outer transformer: head_dim 64, embed_dim % 64 == 0, num_classes % 8 == 0, patch % 8 == 0")
        di, Hi = cfg.inner_embed_dim, cfg.inner_num_heads
        if di % Hi or di // Hi > HDP or di % 8 or cfg.n_pixels > 32 or (cfg.n_pixels * di) % 64:
            raise ValueError("inner transformer: head width <= 16, inner_embed_dim % 8 == 0, <= 32 pixel tokens")
        if not torch.cuda.is_available():
            raise RuntimeError("TNTEngine needs a GPU: there is no CPU path")
        self.L = _lib.load()
        self.cfg = cfg
        self.B = int(batch)
        self.dev = torch.device(device)
        self.rp = int(round_like_reference)
        self._init_cu_budget()
        self.layout = lay = TNTLayout(cfg)
        do, Fo, C, N, NL, n, npx = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers, cfg.n_patches, cfg.n_pixels
        Fi, dap = lay.Fi, lay.dap
        # 16 pixel tokens x 4 heads = one 64-lane wave per sequence; other geometries go through the tiled attention kernels
        self.seq16 = cfg.n_pixels == 16 and Hi == 4 and self.opt.tnt_seq16
        self.Kpi = Kpi = _align(di, 32)            # GEMM K for operands of width di
        self.Kpx = Kpx = _align(lay.pix_in, 64)    # pixel-embedding operand pitch
        self.M = Mo = self.B * N                   # patch-stream rows (name shared with the ViT engine)
        self.Mi = Mi = self.B * n * npx            # pixel-stream rows
        self.Ms = Ms = self.B * n                  # pixel sequences = Inner2Outer rows
        self.Cp = _align(C, 64)
        z = lambda *s, dt=f32: torch.zeros(*s, dtype=dt, device=self.dev)  # noqa: E731
        e = lambda *s, dt=f32: torch.empty(*s, dtype=dt, device=self.dev)  # noqa: E731
        self._init_flat_buffers()
        # bf16 operand copies; the ones that multiply a width-di operand are zero-padded along K (allocated zeroed, the cast
        # writes the real columns only)
        self.w = {
            "iWqkv_n": e(NL, di, 3 * dap, dt=bf16), "iWqkv_t": z(NL, 3 * dap, Kpi, dt=bf16),
            "iWo_n": z(NL, dap, Kpi, dt=bf16), "iWo_t": e(NL, di, dap, dt=bf16),
            "iW1_n": e(NL, di, Fi, dt=bf16), "iW1_t": z(NL, Fi, Kpi, dt=bf16),
            "iW2_n": z(NL, Fi, Kpi, dt=bf16), "iW2_t": e(NL, di, Fi, dt=bf16),
            "Wio_n": e(NL, npx * di, do, dt=bf16), "Wio_t": e(NL, do, npx * di, dt=bf16),
            "Wqkv_n": e(NL, do, 3 * do, dt=bf16), "Wqkv_t": e(NL, 3 * do, do, dt=bf16),
            "Wo_n": e(NL, do, do, dt=bf16), "Wo_t": e(NL, do, do, dt=bf16),
            "W1_n": e(NL, do, Fo, dt=bf16), "W1_t": e(NL, Fo, do, dt=bf16),
            "W2_n": e(NL, Fo, do, dt=bf16), "W2_t": e(NL, do, Fo, dt=bf16),
            "Wpe_t": e(do, cfg.patch_dim, dt=bf16), "Wpx_t": z(di, Kpx, dt=bf16),
            "Wh_t": e(C, do, dt=bf16), "Wh_n": z(do, self.Cp, dt=bf16),
        }
        # ---- activations saved for backward.  Width-di bf16 operands get one spare row: the K round-up reads past the last row
        # only through the buffer descriptor (zeros), the spare row keeps plain loads of other kernels in bounds as well.
        self.pix = z(Mi, Kpx, dt=bf16)                                  # gathered pixel features (48 of 64 columns)
        self.xi = [e(Mi, di) for _ in range(NL + 1)]                    # pixel stream entering layer l
        self.ximid = [e(Mi, di) for _ in range(NL)]
        self.ih1 = [e(Mi, di, dt=bf16) for _ in range(NL)]
        self.ih2 = [e(Mi, di, dt=bf16) for _ in range(NL)]
        self.iqkv = [e(Mi, 3 * dap, dt=bf16) for _ in range(NL)]
        self.io = [e(Mi, dap, dt=bf16) for _ in range(NL)]
        self.iu = [e(Mi, Fi, dt=bf16) for _ in range(NL)]
        self.ia = [e(Mi, Fi, dt=bf16) for _ in range(NL)]
        self.istats = [e(4, Mi) for _ in range(NL)]
        self.ilse = [e(Ms, cfg.inner_num_heads, npx) for _ in range(NL)]
        self.iob = [e(Ms, npx * di, dt=bf16) for _ in range(NL)]        # bf16 copy of the block's pixel output (Inner2Outer operand)
        self.x = [e(Mo, do) for _ in range(NL + 1)]                     # patch stream entering layer l
        self.t = [e(Mo, do) for _ in range(NL)]                         # patch stream + Inner2Outer (the outer LayerNorm's input)
        self.xmid = [e(Mo, do) for _ in range(NL)]
        self.h1 = [e(Mo, do, dt=bf16) for _ in range(NL)]
        self.h2 = [e(Mo, do, dt=bf16) for _ in range(NL)]
        self.qkv = [e(Mo, 3 * do, dt=bf16) for _ in range(NL)]
        self.o = [e(Mo, do, dt=bf16) for _ in range(NL)]
        self.u = [e(Mo, Fo, dt=bf16) for _ in range(NL)]
        self.a = [e(Mo, Fo, dt=bf16) for _ in range(NL)]
        self.stats = [e(4, Mo) for _ in range(NL)]
        self.lse = [e(self.B, cfg.num_heads, N) for _ in range(NL)]
        self.yb = e(Ms, do, dt=bf16)                                    # Inner2Outer projection (scratch)
        self.zcls = e(self.B, do, dt=bf16)
        # ---- backward scratch
        depth = max(2, self.opt.ring_depth)
        self.dres = e(Mo, do)                                           # patch-stream cotangent
        self.dres_b_ring = [e(Mo, do, dt=bf16) for _ in range(2 * depth)]
        self.dres_b = self.dres_b_ring[0]
        self.dresi = e(Mi, di)                                          # pixel-stream cotangent
        self.dresi_b_ring = [e(Mi, di, dt=bf16) for _ in range(2 * depth + 1)]
        self.dt = e(Mo, do)
        self.dy_ring = [e(Ms, do, dt=bf16) for _ in range(depth)]
        self.d_u_ring = [e(Mo, Fo, dt=bf16) for _ in range(depth)]
        self.dqkv_ring = [e(Mo, 3 * do, dt=bf16) for _ in range(depth)]
        self.d_h = e(Mo, do, dt=bf16)
        self.d_o = e(Mo, do, dt=bf16)
        self.id_u_ring = [e(Mi, Fi, dt=bf16) for _ in range(depth)]
        self.idqkv_ring = [e(Mi, 3 * dap, dt=bf16) for _ in range(depth)]
        self.id_h = e(Mi, di, dt=bf16)
        self.id_o = e(Mi, dap, dt=bf16)
        self.colsum_slab = e(max(1, self.L.savit_gemm_colsum_rows_cus(Mo, Fo, do, 0, self.cu_budget if self.reserved_cus else 0)), Fo)
        self.icolsum_slab = e(max(1, self.L.savit_gemm_colsum_rows_cus(Mi, Fi, Kpi, 0, self.cu_budget if self.reserved_cus else 0)), Fi)
        self.d_z = e(self.B, do, dt=bf16)
        ws = max(self.L.savit_layernorm_bwd_workspace_bytes(Mo, do), self.L.savit_layernorm_bwd_workspace_bytes(Mi, di))
        self.ln_ws = torch.empty(max(int(ws), 16), dtype=torch.uint8, device=self.dev)
        self._init_step_state()

    # ------------------------------------------------------------------------------------ parameters
    def init_params(self, seed: int = 0):
        """Reference initialisers: lecun-normal Dense kernels, zero biases, zero cls (tnt.py:164), normal(0.02) position
        embeddings on both streams, LayerNorm 1 / 0, ZERO head kernel (tnt.py:191)."""
        g = torch.Generator(device="cpu").manual_seed(int(seed))
        self.params.zero_()
        tree = self.param_tree()["params"]

        def lecun(t: torch.Tensor, fan_in: int):
            std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
            w = torch.empty(tuple(t.shape), dtype=f32)
            torch.nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=g)
            t.copy_(w)

        def dense(dd):
            k = dd["kernel"]
            lecun(k, k.shape[0])

        def attn(a, width):
            for nme in ("queries", "keys", "values"):
                lecun(a[nme]["kernel"], width)
            lecun(a["DenseGeneral_0"]["kernel"], width)  # fan-in = heads * head width

        dense(tree["PixelEmbedBlock_0"]["Dense_0"])
        dense(tree["PatchEmbedBlock_0"]["Dense_0"])
        tree["AddAbsPosEmbed_0"]["pos_embed"].copy_(torch.randn(tuple(tree["AddAbsPosEmbed_0"]["pos_embed"].shape), generator=g) * 0.02)
        tree["AddAbsPosEmbed_1"]["pos_embed"].copy_(torch.randn(tuple(tree["AddAbsPosEmbed_1"]["pos_embed"].shape), generator=g) * 0.02)
        cfg = self.cfg
        for l in range(cfg.num_layers):
            b = tree["Encoder_0"][f"EncoderBlock_{l}"]
            for i in range(4):
                b[f"LayerNorm_{i}"]["scale"].fill_(1.0)
            attn(b["SelfAttentionBlock_0"], cfg.inner_embed_dim)
            attn(b["SelfAttentionBlock_1"], cfg.embed_dim)
            for ff in ("FFBlock_0", "FFBlock_1"):
                dense(b[ff]["Dense_0"])
                dense(b[ff]["Dense_1"])
            dense(b["Inner2OuterBlock_0"]["Dense_0"])
        self.weights_stale = True

    # ------------------------------------------------------------------------------------ plans
    def _build_cast_plan(self) -> _Plan:
        P, L, lay, cfg = _Plan(), self.L, self.layout, self.cfg
        do, Fo, C, NL, di, npx = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.num_layers, cfg.inner_embed_dim, cfg.n_pixels
        Fi, dap, Kpi = lay.Fi, lay.dap, self.Kpi
        ls = lay.layer_stride

        def cast(name, R, Cc, n_ld, t_ld):
            wn, wt = self.w[name + "_n"], self.w[name + "_t"]
            P.add(L.savit_cast_transpose_bf16, (self._off_ptr(self.params, f"l0.{name}"), ls, NL, R, Cc, wn.data_ptr(), wn.shape[1] * wn.shape[2],
                                                n_ld, wt.data_ptr(), wt.shape[1] * wt.shape[2], t_ld), f"cast {name}")

        cast("iWqkv", di, 3 * dap, 3 * dap, Kpi)
        cast("iWo", dap, di, Kpi, dap)
        cast("iW1", di, Fi, Fi, Kpi)
        cast("iW2", Fi, di, Kpi, Fi)
        cast("Wio", npx * di, do, do, npx * di)
        cast("Wqkv", do, 3 * do, 3 * do, do)
        cast("Wo", do, do, do, do)
        cast("W1", do, Fo, Fo, do)
        cast("W2", Fo, do, do, Fo)
        P.add(L.savit_cast_transpose_bf16, (self._off_ptr(self.params, "Wpe"), 0, 1, cfg.patch_dim, do, None, 0, do,
                                            self.w["Wpe_t"].data_ptr(), 0, cfg.patch_dim), "cast Wpe")
        P.add(L.savit_cast_transpose_bf16, (self._off_ptr(self.params, "Wpx"), 0, 1, lay.pix_in, di, None, 0, di,
                                            self.w["Wpx_t"].data_ptr(), 0, self.Kpx), "cast Wpx")
        P.add(L.savit_cast_transpose_bf16, (self._off_ptr(self.params, "Wh"), 0, 1, do, C, self.w["Wh_n"].data_ptr(), 0, self.Cp,
                                            self.w["Wh_t"].data_ptr(), 0, do), "cast Wh")
        return P

    def _build_fwd_plan(self) -> _Plan:
        P, L, cfg, lay = _Plan(), self.L, self.cfg, self.layout
        do, Fo, C, N, NL, Ho, B = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers, cfg.num_heads, self.B
        di, Hi, npx, n = cfg.inner_embed_dim, cfg.inner_num_heads, cfg.n_pixels, cfg.n_patches
        Fi, dap, Kpi, Kpx, Mo, Mi, Ms = lay.Fi, lay.dap, self.Kpi, self.Kpx, self.M, self.Mi, self.Ms
        pp = lambda nme: self._off_ptr(self.params, nme)  # noqa: E731
        x, xi = self.x, self.xi
        # ---- embeddings (tnt.py:151-171)
        P.add(L.savit_tnt_pixel_gather, (self._img_buf.data_ptr(), self.pix.data_ptr(), B, cfg.img_size, cfg.patch, cfg.transformed_patch, 3, Kpx),
              "pixel_gather")
        self._gemm(P, "pixel_embed", A=self.pix.data_ptr(), Bt=self.w["Wpx_t"].data_ptr(), C=xi[0].data_ptr(), bias=pp("bpx"), M=Mi, N=di, K=Kpx,
                   lda=Kpx, ldb=Kpx, ldc=di, epilogue=_lib.EPI_F32, round_out_bf16=self.rp)
        P.add(L.savit_add_rows_periodic, (xi[0].data_ptr(), pp("ppos"), Mi, npx, di), "pixel_pos")
        self._gemm(P, "patch_embed", A=self._img_buf.data_ptr(), Bt=self.w["Wpe_t"].data_ptr(), C=x[0].data_ptr(), bias=pp("bpa"), aux=pp("pos"),
                   M=B * n, N=do, K=cfg.patch_dim, lda=0, ldb=cfg.patch_dim, ldc=do, ldaux=do, epilogue=_lib.EPI_PATCH, img_size=cfg.img_size,
                   patch=cfg.patch, tokens=N, token_offset=1)
        P.add(L.savit_cls_pos_rows, (pp("cls"), pp("pos"), x[0].data_ptr(), B, N * do, do), "cls_rows")
        ialpha, oalpha = 1.0 / math.sqrt(di // Hi), 1.0 / math.sqrt(cfg.head_dim)
        for l in range(NL):
            p = f"l{l}."
            w = lambda nme, l=l: self.w[nme][l].data_ptr()  # noqa: E731
            ist, st = self.istats[l], self.stats[l]
            # inner block on the pixel stream (tnt.py:68-80)
            P.add(L.savit_layernorm_fwd, (xi[l].data_ptr(), pp(p + "iln1_g"), pp(p + "iln1_b"), self.ih1[l].data_ptr(), ist[0].data_ptr(),
                                          ist[1].data_ptr(), Mi, di, di, 1e-6, self.rp), p + "iln1")
            self._gemm(P, p + "iqkv", A=self.ih1[l].data_ptr(), Bt=w("iWqkv_t"), C=self.iqkv[l].data_ptr(), M=Mi, N=3 * dap, K=Kpi, lda=di, ldb=Kpi,
                       ldc=3 * dap, epilogue=_lib.EPI_BF16, alpha=ialpha, alpha_cols=dap)
            if self.seq16:  # one wave per pixel sequence (seq16_attention.hip)
                P.add(L.savit_seq16_attention_fwd, (self.iqkv[l].data_ptr(), self.io[l].data_ptr(), Ms, npx, Hi, HDP, 3 * dap), p + "iattn")
            else:
                P.add(L.savit_attention_fwd, (self.iqkv[l].data_ptr(), self.io[l].data_ptr(), self.ilse[l].data_ptr(), Ms, npx, Hi, HDP, 3 * dap),
                      p + "iattn")
            self._gemm(P, p + "iproj", A=self.io[l].data_ptr(), Bt=w("iWo_t"), C=self.ximid[l].data_ptr(), aux=xi[l].data_ptr(), M=Mi, N=di, K=dap,
                       lda=dap, ldb=dap, ldc=di, ldaux=di, epilogue=_lib.EPI_RESID)
            P.add(L.savit_layernorm_fwd, (self.ximid[l].data_ptr(), pp(p + "iln2_g"), pp(p + "iln2_b"), self.ih2[l].data_ptr(), ist[2].data_ptr(),
                                          ist[3].data_ptr(), Mi, di, di, 1e-6, self.rp), p + "iln2")
            self._gemm(P, p + "ifc1", A=self.ih2[l].data_ptr(), Bt=w("iW1_t"), C=self.iu[l].data_ptr(), C2=self.ia[l].data_ptr(), bias=pp(p + "ib1"),
                       M=Mi, N=Fi, K=Kpi, lda=di, ldb=Kpi, ldc=Fi, epilogue=_lib.EPI_BIAS_GELU)
            self._gemm(P, p + "ifc2", A=self.ia[l].data_ptr(), Bt=w("iW2_t"), C=xi[l + 1].data_ptr(), bias=pp(p + "ib2"), aux=self.ximid[l].data_ptr(),
                       M=Mi, N=di, K=Fi, lda=Fi, ldb=Fi, ldc=di, ldaux=di, epilogue=_lib.EPI_RESID)
            # Inner2Outer (tnt.py:40-51): [B*n, 16*di] -> do, zero row for cls, add to the patch stream
            P.add(L.savit_cast_bf16, (xi[l + 1].data_ptr(), self.iob[l].data_ptr(), Mi * di), p + "io.cast")
            self._gemm(P, p + "io.fc", A=self.iob[l].data_ptr(), Bt=w("Wio_t"), C=self.yb.data_ptr(), bias=pp(p + "bio"), M=Ms, N=do, K=npx * di,
                       lda=npx * di, ldb=npx * di, ldc=do, epilogue=_lib.EPI_BF16)
            P.add(L.savit_tnt_inner2outer_add, (x[l].data_ptr(), self.yb.data_ptr(), self.t[l].data_ptr(), B, N, do), p + "io.add")
            # outer block on the patch stream (tnt.py:84-91): attention reads LN(t), the residual adds patch_inputs
            P.add(L.savit_layernorm_fwd, (self.t[l].data_ptr(), pp(p + "ln1_g"), pp(p + "ln1_b"), self.h1[l].data_ptr(), st[0].data_ptr(),
                                          st[1].data_ptr(), Mo, do, do, 1e-6, self.rp), p + "ln1")
            self._gemm(P, p + "qkv", A=self.h1[l].data_ptr(), Bt=w("Wqkv_t"), C=self.qkv[l].data_ptr(), M=Mo, N=3 * do, K=do, lda=do, ldb=do,
                       ldc=3 * do, epilogue=_lib.EPI_BF16, alpha=oalpha, alpha_cols=do)
            P.add(L.savit_attention_fwd, (self.qkv[l].data_ptr(), self.o[l].data_ptr(), self.lse[l].data_ptr(), B, N, Ho, cfg.head_dim, 3 * do),
                  p + "attn")
            self._gemm(P, p + "proj", A=self.o[l].data_ptr(), Bt=w("Wo_t"), C=self.xmid[l].data_ptr(), aux=x[l].data_ptr(), M=Mo, N=do, K=do,
                       lda=do, ldb=do, ldc=do, ldaux=do, epilogue=_lib.EPI_RESID)
            P.add(L.savit_layernorm_fwd, (self.xmid[l].data_ptr(), pp(p + "ln2_g"), pp(p + "ln2_b"), self.h2[l].data_ptr(), st[2].data_ptr(),
                                          st[3].data_ptr(), Mo, do, do, 1e-6, self.rp), p + "ln2")
            self._gemm(P, p + "fc1", A=self.h2[l].data_ptr(), Bt=w("W1_t"), C=self.u[l].data_ptr(), C2=self.a[l].data_ptr(), bias=pp(p + "b1"),
                       M=Mo, N=Fo, K=do, lda=do, ldb=do, ldc=Fo, epilogue=_lib.EPI_BIAS_GELU)
            self._gemm(P, p + "fc2", A=self.a[l].data_ptr(), Bt=w("W2_t"), C=x[l + 1].data_ptr(), bias=pp(p + "b2"), aux=self.xmid[l].data_ptr(),
                       M=Mo, N=do, K=Fo, lda=Fo, ldb=Fo, ldc=do, ldaux=do, epilogue=_lib.EPI_RESID)
        # head on the cls row, no LayerNorm (tnt.py:187-193)
        P.add(L.savit_gather_rows_bf16, (x[NL].data_ptr(), N * do, self.zcls.data_ptr(), B, do), "cls_gather")
        self._gemm(P, "head", A=self.zcls.data_ptr(), Bt=self.w["Wh_t"].data_ptr(), C=self.logits.data_ptr(), bias=pp("bh"), M=B, N=C, K=do,
                   lda=do, ldb=do, ldc=C, epilogue=_lib.EPI_F32, round_out_bf16=self.rp)
        return P

    def _record_bwd_plan(self) -> _Plan:
        P, L, cfg, lay = _Plan(), self.L, self.cfg, self.layout
        do, Fo, C, N, NL, Ho, B = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers, cfg.num_heads, self.B
        di, Hi, npx, n = cfg.inner_embed_dim, cfg.inner_num_heads, cfg.n_pixels, cfg.n_patches
        Fi, dap, Kpi, Kpx, Mo, Mi, Ms = lay.Fi, lay.dap, self.Kpi, self.Kpx, self.M, self.Mi, self.Ms
        pp = lambda nme: self._off_ptr(self.params, nme)  # noqa: E731
        gp = lambda nme: self._off_ptr(self.grads, nme)  # noqa: E731
        ws, wsb = self.ln_ws.data_ptr(), self.ln_ws.numel()

        # The pixel-stream weight gradients reduce B*n*16 rows into one or two 128x128 output tiles: the library's default of 24
        # K-splits leaves them at 166 us each (4 per layer: the longest item of the step); SAVIT_TNT_INNER_SPLITS sizes them.
        inner_splits = int(self.opt.tnt_inner_splits)

        def wgrad(label, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, patch=(0, 0, 0, 0)):
            sp = self._wgrad_splits(Kin, Nout, patch[0])
            if Mr == Mi and not patch[0]:
                tiles = -(-Kin // 128) * -(-Nout // 128)
                sp = max(1, inner_splits // tiles)
            self._add_wgrad(P, label, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, sp, patch)

        ring, ri = [t.data_ptr() for t in self.dres_b_ring], 0
        iring, iri = [t.data_ptr() for t in self.dresi_b_ring], 0
        dres, dresi, dt = self.dres.data_ptr(), self.dresi.data_ptr(), self.dt.data_ptr()
        # ---- head: dWh, d z_cls scattered into the cls rows of the (zeroed) patch-stream cotangent
        wgrad("head.wgrad", self.zcls.data_ptr(), self.dlogits.data_ptr(), gp("Wh"), B, do, C, do, self.Cp, C)
        self._gemm(P, "head.dgrad", A=self.dlogits.data_ptr(), Bt=self.w["Wh_n"].data_ptr(), C=self.d_z.data_ptr(), M=B, N=do, K=self.Cp,
                   lda=self.Cp, ldb=self.Cp, ldc=do, epilogue=_lib.EPI_BF16)
        P.add(L.savit_scatter_rows, (self.d_z.data_ptr(), dres, None, N * do, B, do), "cls_scatter")
        for l in range(NL - 1, -1, -1):
            p = f"l{l}."
            w = lambda nme, l=l: self.w[nme][l].data_ptr()  # noqa: E731
            ist, st = self.istats[l], self.stats[l]
            k = l % len(self.d_u_ring)
            d_u, dqkv, dy = self.d_u_ring[k].data_ptr(), self.dqkv_ring[k].data_ptr(), self.dy_ring[k].data_ptr()
            id_u, idqkv = self.id_u_ring[k].data_ptr(), self.idqkv_ring[k].data_ptr()
            # ===== outer block.  dres = d patches[l+1]: completed by the previous split (or the head scatter), not by a LayerNorm
            # backward, so its bf16 copy and its column sums (= the gradient of this layer's second FF bias) come from one pass here
            P.add(L.savit_cast_colsum, (dres, ring[ri], gp(p + "b2"), Mo, do), p + "b2.grad", writes=(ring[ri],))
            wgrad(p + "W2.wgrad", self.a[l].data_ptr(), ring[ri], gp(p + "W2"), Mo, Fo, do, Fo, do, do)
            self._gemm(P, p + "fc2.dgrad", writes=(d_u,), A=ring[ri], Bt=w("W2_n"), C=d_u, aux=self.u[l].data_ptr(),
                       colsum=self.colsum_slab.data_ptr(), colsum_rows=self.colsum_slab.shape[0], M=Mo, N=Fo, K=do, lda=do, ldb=do, ldc=Fo,
                       ldaux=Fo, epilogue=_lib.EPI_DGELU)
            P.add(L.savit_colsum_finalize, (self.colsum_slab.data_ptr(), self.colsum_slab.shape[0], Fo, gp(p + "b1"), 1), p + "b1.grad")
            wgrad(p + "W1.wgrad", self.h2[l].data_ptr(), d_u, gp(p + "W1"), Mo, do, Fo, do, Fo, Fo)
            self._gemm(P, p + "fc1.dgrad", A=d_u, Bt=w("W1_n"), C=self.d_h.data_ptr(), M=Mo, N=do, K=Fo, lda=Fo, ldb=Fo, ldc=do,
                       epilogue=_lib.EPI_BF16)
            ri = (ri + 1) % len(ring)
            P.add(L.savit_layernorm_bwd, (self.d_h.data_ptr(), self.xmid[l].data_ptr(), pp(p + "ln2_g"), st[2].data_ptr(), st[3].data_ptr(), dres,
                                          dres, ring[ri], gp(p + "ln2_g"), gp(p + "ln2_b"), None, Mo, do, do, do, self.rp, ws, wsb),
                  p + "ln2.bwd", writes=(ring[ri],))
            wgrad(p + "Wo.wgrad", self.o[l].data_ptr(), ring[ri], gp(p + "Wo"), Mo, do, do, do, do, do)
            self._gemm(P, p + "proj.dgrad", A=ring[ri], Bt=w("Wo_n"), C=self.d_o.data_ptr(), M=Mo, N=do, K=do, lda=do, ldb=do, ldc=do,
                       epilogue=_lib.EPI_BF16)
            P.add(L.savit_attention_bwd, (self.qkv[l].data_ptr(), self.o[l].data_ptr(), self.d_o.data_ptr(), self.lse[l].data_ptr(), dqkv, B, N,
                                          Ho, cfg.head_dim, 3 * do, 1.0 / math.sqrt(cfg.head_dim)), p + "attn.bwd", writes=(dqkv,))
            wgrad(p + "Wqkv.wgrad", self.h1[l].data_ptr(), dqkv, gp(p + "Wqkv"), Mo, do, 3 * do, do, 3 * do, 3 * do)
            self._gemm(P, p + "qkv.dgrad", A=dqkv, Bt=w("Wqkv_n"), C=self.d_h.data_ptr(), M=Mo, N=do, K=3 * do, lda=3 * do, ldb=3 * do,
                       ldc=do, epilogue=_lib.EPI_BF16)
            # cotangent of t = patches + pad(Inner2Outer): through the LayerNorm only (the residual went to patch_inputs, tnt.py:86)
            P.add(L.savit_layernorm_bwd, (self.d_h.data_ptr(), self.t[l].data_ptr(), pp(p + "ln1_g"), st[0].data_ptr(), st[1].data_ptr(), None,
                                          dt, None, gp(p + "ln1_g"), gp(p + "ln1_b"), None, Mo, do, do, do, self.rp, ws, wsb), p + "ln1o.bwd")
            P.add(L.savit_tnt_inner2outer_split, (dt, dres, dy, gp(p + "bio"), B, N, do), p + "io.split", writes=(dy,))
            wgrad(p + "Wio.wgrad", self.iob[l].data_ptr(), dy, gp(p + "Wio"), Ms, npx * di, do, npx * di, do, do)
            # ===== inner block.  d pixels[l+1] += Inner2Outer input gradient (viewed [B*n, 16*di]); then bf16 copy + column sums
            self._gemm(P, p + "io.dgrad", A=dy, Bt=w("Wio_n"), C=dresi, aux=dresi, M=Ms, N=npx * di, K=do, lda=do, ldb=do, ldc=npx * di,
                       ldaux=npx * di, epilogue=_lib.EPI_RESID)
            P.add(L.savit_cast_colsum, (dresi, iring[iri], gp(p + "ib2"), Mi, di), p + "ib2.grad", writes=(iring[iri],))
            wgrad(p + "iW2.wgrad", self.ia[l].data_ptr(), iring[iri], gp(p + "iW2"), Mi, Fi, di, Fi, di, di)
            self._gemm(P, p + "ifc2.dgrad", writes=(id_u,), A=iring[iri], Bt=w("iW2_n"), C=id_u, aux=self.iu[l].data_ptr(),
                       colsum=self.icolsum_slab.data_ptr(), colsum_rows=self.icolsum_slab.shape[0], M=Mi, N=Fi, K=Kpi, lda=di, ldb=Kpi, ldc=Fi,
                       ldaux=Fi, epilogue=_lib.EPI_DGELU)
            P.add(L.savit_colsum_finalize, (self.icolsum_slab.data_ptr(), self.icolsum_slab.shape[0], Fi, gp(p + "ib1"), 1), p + "ib1.grad")
            wgrad(p + "iW1.wgrad", self.ih2[l].data_ptr(), id_u, gp(p + "iW1"), Mi, di, Fi, di, Fi, Fi)
            self._gemm(P, p + "ifc1.dgrad", A=id_u, Bt=w("iW1_n"), C=self.id_h.data_ptr(), M=Mi, N=di, K=Fi, lda=Fi, ldb=Fi, ldc=di,
                       epilogue=_lib.EPI_BF16)
            iri = (iri + 1) % len(iring)
            P.add(L.savit_layernorm_bwd, (self.id_h.data_ptr(), self.ximid[l].data_ptr(), pp(p + "iln2_g"), ist[2].data_ptr(), ist[3].data_ptr(),
                                          dresi, dresi, iring[iri], gp(p + "iln2_g"), gp(p + "iln2_b"), None, Mi, di, di, di, self.rp, ws, wsb),
                  p + "iln2.bwd", writes=(iring[iri],))
            wgrad(p + "iWo.wgrad", self.io[l].data_ptr(), iring[iri], gp(p + "iWo"), Mi, dap, di, dap, di, di)
            self._gemm(P, p + "iproj.dgrad", A=iring[iri], Bt=w("iWo_n"), C=self.id_o.data_ptr(), M=Mi, N=dap, K=Kpi, lda=di, ldb=Kpi, ldc=dap,
                       epilogue=_lib.EPI_BF16)
            if self.seq16:
                P.add(L.savit_seq16_attention_bwd, (self.iqkv[l].data_ptr(), self.id_o.data_ptr(), idqkv, Ms, npx, Hi, HDP, 3 * dap,
                                                    1.0 / math.sqrt(di // Hi)), p + "iattn.bwd", writes=(idqkv,))
            else:
                P.add(L.savit_attention_bwd, (self.iqkv[l].data_ptr(), self.io[l].data_ptr(), self.id_o.data_ptr(), self.ilse[l].data_ptr(), idqkv,
                                              Ms, npx, Hi, HDP, 3 * dap, 1.0 / math.sqrt(di // Hi)), p + "iattn.bwd", writes=(idqkv,))
            wgrad(p + "iWqkv.wgrad", self.ih1[l].data_ptr(), idqkv, gp(p + "iWqkv"), Mi, di, 3 * dap, di, 3 * dap, 3 * dap)
            self._gemm(P, p + "iqkv.dgrad", A=idqkv, Bt=w("iWqkv_n"), C=self.id_h.data_ptr(), M=Mi, N=di, K=3 * dap, lda=3 * dap, ldb=3 * dap,
                       ldc=di, epilogue=_lib.EPI_BF16)
            iri = (iri + 1) % len(iring)
            # last launch that writes layer l's gradients: the label the DDP bucket hooks wait for
            P.add(L.savit_layernorm_bwd, (self.id_h.data_ptr(), self.xi[l].data_ptr(), pp(p + "iln1_g"), ist[0].data_ptr(), ist[1].data_ptr(),
                                          dresi, dresi, iring[iri], gp(p + "iln1_g"), gp(p + "iln1_b"), None, Mi, di, di, di, self.rp, ws, wsb),
                  p + "ln1.bwd", writes=(iring[iri],))
        # ---- embeddings.  pixel stream: position embedding = column sums of the [B*n, 16*di] view; bias = its sum over the 16 tokens
        P.add(L.savit_cast_colsum, (dresi, None, gp("ppos"), Ms, npx * di), "ppos.grad")
        P.add(L.savit_colsum_finalize, (gp("ppos"), npx, di, gp("bpx"), 1), "bpx.grad")
        wgrad("Wpx.wgrad", self.pix.data_ptr(), iring[iri], gp("Wpx"), Mi, lay.pix_in, di, Kpx, di, di)
        # patch stream: pos / cls as in ViT; the patch-embedding bias = sum of the position gradient over the patch rows
        P.add(L.savit_cast_colsum, (dres, ring[ri], None, Mo, do), "dx0.cast", writes=(ring[ri],))
        P.add(L.savit_pos_cls_grad, (dres, gp("pos"), gp("cls"), B, N, do, 1), "pos_cls.grad")
        P.add(L.savit_colsum_finalize, (gp("pos") + do * 4, n, do, gp("bpa"), 1), "bpa.grad")
        wgrad("Wpe.wgrad", self._img_buf.data_ptr(), ring[ri], gp("Wpe"), B * n, cfg.patch_dim, do, 0, do, do,
              patch=(cfg.patch, cfg.img_size, N, 1))
        finalize_wgrad_ws(self, P)
        return P

    def _backward_from_dlogits(self):
        self._zero("zero.dresi", self.dresi)  # the last layer's pixel output feeds Inner2Outer only: its cotangent starts from zero
        super()._backward_from_dlogits()

    def activation_bytes(self) -> int:
        tot = 0
        for group in (self.xi, self.ximid, self.ih1, self.ih2, self.iqkv, self.io, self.iu, self.ia, self.istats, self.ilse, self.iob, self.x, self.t,
                      self.xmid, self.h1, self.h2, self.qkv, self.o, self.u, self.a, self.stats, self.lse):
            tot += sum(t.numel() * t.element_size() for t in group)
        return tot
