"""CaiT training engine (models/cait.py:140-183 of the reference): patch-embed -> L x [LN -> talking-heads SA -> LayerScale ->
StochasticDepth -> +res ; LN -> FF -> LayerScale -> StochasticDepth -> +res] -> 2 x class-attention blocks that update only the
cls token -> LN -> head.  Same design as engine.ViTEngine: flat fp32 parameter / gradient buffers (layer-major), bf16 MFMA operand
copies in both layouts, prebuilt launch plans, every tensor operation a C-ABI kernel.

Pieces specific to CaiT
  * talking-heads attention (attention.py:44-52): savit_th_attention_fwd/bwd with S and P' as bf16 [B,H,N,Np] HBM tensors, or (opt-in,
    SAVIT_TH_FUSED=1) savit_th_fused_attention_fwd/bwd with S and P' in LDS;
  * LayerScale + stochastic depth (cait.py:36-40,47-52): fused into the residual GEMM epilogue (colscale / rowscale), which also
    stores the bf16 branch so that savit_layerscale_bwd can form d(layerscale) and the branch cotangent;
  * class attention (cait.py:96-122): LayerNorm over [cls; x] is two row-mapped LN calls into one [B*(N+1), d] operand (no fp32
    concat); q is projected for the cls rows only, K|V for all rows, through ONE fused Wqkv whose q rows see a zero cotangent
    except at the cls rows; savit_class_attention_fwd/bwd does the 1 x (N+1) softmax per (batch, head);
  * stochastic-depth masks are drawn per step with torch's generator on the GPU (the JAX 'stochastic_depth' rng stream cannot be
    reproduced) or supplied by the caller (tests): rowscale = floor(keep + U) / keep per sample and branch.
The reference runs the SA encoder in fp32 because cait.py:147-154 forgets to forward dtype (defect B8); this engine runs all of
CaiT in bf16 like the ViT path and is checked against the fp32 oracle.
"""
from __future__ import annotations

import ctypes
import math
import os as _os
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as _lib
from .config import ModelConfig
from .engine import (ViTEngine, WgradQueue, _Plan, _align, _copy_tree, add_wgrad, add_wgrad_group, finalize_first_touch, finalize_wgrad_ws,
                     wgrad_group_tile)
from .options import EngineOptions
from .timing import timed_call

bf16 = torch.bfloat16
f32 = torch.float32


class CaiTLayout:
    def __init__(self, cfg: ModelConfig):
        self.cfg = cfg
        d, F, C, N, H = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.n_patches, cfg.num_heads
        self.off: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        cur = 0

        def add(name, shape):
            nonlocal cur
            n = 1
            for s in shape:
                n *= s
            self.off[name] = (cur, tuple(shape))
            cur += _align(n, 4)

        add("Wpe", (cfg.patch_dim, d))
        add("pos", (N, d))
        cur = _align(cur, 64)
        self.layer_start: List[int] = []
        for l in range(cfg.num_layers):
            self.layer_start.append(cur)
            for nm, shp in (("ln1_g", (d,)), ("ln1_b", (d,)), ("Wqkv", (d, 3 * d)), ("T1", (H, H)), ("T2", (H, H)), ("Wo", (d, d)), ("ls1", (d,)),
                            ("ln2_g", (d,)), ("ln2_b", (d,)), ("W1", (d, F)), ("b1", (F,)), ("W2", (F, d)), ("b2", (d,)), ("ls2", (d,))):
                add(f"l{l}.{nm}", shp)
            cur = _align(cur, 64)
        self.layer_stride = (self.layer_start[1] - self.layer_start[0]) if cfg.num_layers > 1 else (cur - self.layer_start[0])
        self.ca_start: List[int] = []
        for c in range(cfg.num_layers_token_only):
            self.ca_start.append(cur)
            for nm, shp in (("ln1_g", (d,)), ("ln1_b", (d,)), ("Wqkv", (d, 3 * d)), ("Wo", (d, d)), ("ls1", (d,)), ("ln2_g", (d,)), ("ln2_b", (d,)),
                            ("W1", (d, F)), ("b1", (F,)), ("W2", (F, d)), ("b2", (d,)), ("ls2", (d,))):
                add(f"c{c}.{nm}", shp)
            cur = _align(cur, 64)
        self.ca_stride = (self.ca_start[1] - self.ca_start[0]) if cfg.num_layers_token_only > 1 else (cur - (self.ca_start[0] if self.ca_start else cur))
        self.final_start = cur
        add("cls", (d,))
        add("lnf_g", (d,))
        add("lnf_b", (d,))
        add("Wh", (d, C))
        add("bh", (C,))
        self.total = _align(cur, 64)

    def view(self, flat, name):
        o, shape = self.off[name]
        n = 1
        for s in shape:
            n *= s
        return flat[o:o + n].view(*shape)

    def flax_tree(self, flat) -> dict:
        """Flax-shaped tree of views (CaiT additions)."""
        cfg = self.cfg
        d, H, hd = cfg.embed_dim, cfg.num_heads, cfg.head_dim
        v = lambda n: self.view(flat, n)  # noqa: E731

        def attn(prefix, talking):
            w = v(prefix + ".Wqkv")
            t = {"queries": {"kernel": w[:, 0:d].unflatten(1, (H, hd))}, "keys": {"kernel": w[:, d:2 * d].unflatten(1, (H, hd))},
                 "values": {"kernel": w[:, 2 * d:3 * d].unflatten(1, (H, hd))}, "DenseGeneral_0": {"kernel": v(prefix + ".Wo").view(H, hd, d)}}
            if talking:
                t["TalkingHeadsBlock_0"] = {"talking_heads_transform": v(prefix + ".T1")}
                t["TalkingHeadsBlock_1"] = {"talking_heads_transform": v(prefix + ".T2")}
            return t

        def block(prefix, attn_name, talking):
            return {"LayerNorm_0": {"scale": v(prefix + ".ln1_g"), "bias": v(prefix + ".ln1_b")}, attn_name: attn(prefix, talking),
                    "LayerScaleBlock_0": {"layerscale": v(prefix + ".ls1")},
                    "LayerNorm_1": {"scale": v(prefix + ".ln2_g"), "bias": v(prefix + ".ln2_b")},
                    "FFBlock_0": {"Dense_0": {"kernel": v(prefix + ".W1"), "bias": v(prefix + ".b1")},
                                  "Dense_1": {"kernel": v(prefix + ".W2"), "bias": v(prefix + ".b2")}},
                    "LayerScaleBlock_1": {"layerscale": v(prefix + ".ls2")}}

        enc = {"AddAbsPosEmbed_0": {"pos_embed": v("pos").view(1, cfg.n_patches, d)}}
        for l in range(cfg.num_layers):
            enc[f"EncoderBlock_{l}"] = block(f"l{l}", "SelfAttentionBlock_0", True)
        p = {"PatchEmbedBlock_0": {"Dense_0": {"kernel": v("Wpe")}}, "Encoder_0": enc, "cls": v("cls").view(1, 1, d),
             "LayerNorm_0": {"scale": v("lnf_g"), "bias": v("lnf_b")}, "Dense_0": {"kernel": v("Wh"), "bias": v("bh")}}
        for c in range(cfg.num_layers_token_only):
            p[f"CAEncoderBlock_{c}"] = block(f"c{c}", "ClassSelfAttentionBlock_0", False)
        return {"params": p}


def stochastic_depth_seed(seed: int, rank: int, step: int) -> int:
    """Seed of the stochastic-depth masks of one (run seed, data-parallel rank, global step): Flax draws them from a per-device,
    per-step 'stochastic_depth' rng (stochastic_depth.py:17-21 under pmap; train.py never passes one - defect B7).  Every rank gets
    its own masks, a resumed run continues with the masks it would have drawn, and --seed changes all of them."""
    x = (int(seed) * 1000003 + int(rank)) * 0x9E3779B97F4A7C15 + int(step) * 0xBF58476D1CE4E5B9
    x &= (1 << 64) - 1
    x ^= x >> 31
    return x & ((1 << 63) - 1)


class CaiTEngine:
    def __init__(self, cfg: ModelConfig, batch: int, device: str = "cuda", round_like_reference: bool = True,
                 th_fused: "bool | None" = None, reserved_cus=None, wgrad_max_lag=None, options=None, **opts):
        self.opt = EngineOptions.resolve(options, th_fused=th_fused, reserved_cus=reserved_cus, wgrad_max_lag=wgrad_max_lag, **opts)
        if cfg.kind != "cait":
            raise ValueError("CaiTEngine needs a CaiT config")
        if cfg.head_dim not in (48, 64) or cfg.num_heads not in (2, 4, 6, 8, 16):
            raise NotImplementedError("talking-heads kernels: head_dim 48/64 and 2/4/6/8/16 heads")
        if cfg.n_patches + 1 > 256:
            raise NotImplementedError("talking-heads / class-attention kernels: at most 255 patches")
        if cfg.embed_dim % 32 or cfg.patch % 8 or cfg.num_classes % 8:
            raise ValueError("embed_dim % 32, patch % 8 and num_classes % 8 must be 0")
        if not torch.cuda.is_available():
            raise RuntimeError("CaiTEngine needs a GPU: there is no CPU path")
        self.L = _lib.load()
        self.cfg, self.B, self.dev, self.rp = cfg, int(batch), torch.device(device), int(round_like_reference)
        self.layout = CaiTLayout(cfg)
        d, F, C, N, NL, NC, H = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.n_patches, cfg.num_layers, cfg.num_layers_token_only, cfg.num_heads
        B = self.B
        self.M, self.Mc = B * N, B * (N + 1)
        self.Cp, self.Np = _align(C, 64), _align(N, 8)
        z = lambda *s, dt=f32: torch.zeros(*s, dtype=dt, device=self.dev)  # noqa: E731
        e = lambda *s, dt=f32: torch.empty(*s, dtype=dt, device=self.dev)  # noqa: E731
        self.params, self.grads = z(self.layout.total), z(self.layout.total)
        self.adam_m = self.adam_v = None
        self.step_count = 0
        self.gnorm = z(36)  # [0]: squared gradient norm; [1:33]: accumulators of the grouped weight-gradient launches (engine.py, round 5)
        self.gnorm_sq = self.gnorm[0:1]
        self.w = {}
        for pre, n in (("", NL), ("c", NC)):
            for nm, R, Cc in (("Wqkv", d, 3 * d), ("Wo", d, d), ("W1", d, F), ("W2", F, d)):
                self.w[pre + nm + "_n"] = e(n, R, Cc, dt=bf16)
                self.w[pre + nm + "_t"] = e(n, Cc, R, dt=bf16)
        self.w["Wpe_t"], self.w["Wh_t"], self.w["Wh_n"] = e(d, cfg.patch_dim, dt=bf16), e(C, d, dt=bf16), z(d, self.Cp, dt=bf16)
        M, Mc = self.M, self.Mc
        # ---- SA activations
        self.x = [e(M, d) for _ in range(NL + 1)]
        self.xmid = [e(M, d) for _ in range(NL)]
        self.h1, self.h2, self.o = ([e(M, d, dt=bf16) for _ in range(NL)] for _ in range(3))
        self.br1, self.br2 = ([e(M, d, dt=bf16) for _ in range(NL)] for _ in range(2))
        self.qkv = [e(M, 3 * d, dt=bf16) for _ in range(NL)]
        # talking-heads attention: the materialising kernels (S and P' [B,H,N,Np] saved per layer for backward), or with
        # th_fused=True (or SAVIT_TH_FUSED=1, read HERE - the library reads no environment) the fused ones (S / P' in LDS, nothing kept
        # per layer; slower on MI355X: see csrc/th_fused.hip).  Default: what the library prefers for this geometry.
        th_fused = self.opt.th_fused
        if th_fused is None:
            th_fused = bool(self.L.savit_th_fused_preferred(N, H, cfg.head_dim))
        self.th_fused = bool(th_fused) and bool(self.L.savit_th_fused_supported(N, H, cfg.head_dim))
        if self.th_fused:
            self.sbuf, self.pbuf = [], []
            self.th_pbuf = e(B, H, N, self.Np, dt=bf16)  # backward scratch: P' (dsbuf below takes dS)
            self.th_vt = torch.empty(max(int(self.L.savit_th_fused_fwd_workspace_bytes(B, N, H, cfg.head_dim)), 16), dtype=torch.uint8, device=self.dev)
        else:
            self.sbuf = [e(B, H, N, self.Np, dt=bf16) for _ in range(NL)]
            self.pbuf = [e(B, H, N, self.Np, dt=bf16) for _ in range(NL)]
        self.u, self.a = ([e(M, F, dt=bf16) for _ in range(NL)] for _ in range(2))
        self.stats = [e(4, M) for _ in range(NL)]
        # ---- CA activations
        self.cls = [e(B, d) for _ in range(NC + 1)]
        self.clsmid = [e(B, d) for _ in range(NC)]
        self.hc = [e(Mc, d, dt=bf16) for _ in range(NC)]
        self.qkvc = [z(Mc, 3 * d, dt=bf16) for _ in range(NC)]
        self.oc, self.cbr1, self.cbr2, self.hc2 = ([e(B, d, dt=bf16) for _ in range(NC)] for _ in range(4))
        self.probs = [e(B, H, N + 1) for _ in range(NC)]
        self.uc, self.ac = ([e(B, F, dt=bf16) for _ in range(NC)] for _ in range(2))
        self.cstat_x = [e(2, M) for _ in range(NC)]
        self.cstat_c = [e(4, B) for _ in range(NC)]
        self.zero_d = z(d)
        self.zcls, self.fstats, self.logits = e(B, d, dt=bf16), e(2, B), e(B, C)
        # ---- stochastic depth scales [(NL + NC) * 2, B]
        self.sd = torch.ones((NL + NC) * 2, B, dtype=f32, device=self.dev)
        self.gen = torch.Generator(device=self.dev).manual_seed(0)
        # ---- backward scratch
        self.dres, self.dres_b = e(M, d), e(M, d, dt=bf16)
        self.d_h, self.d_o = e(M, d, dt=bf16), e(M, d, dt=bf16)
        # The SA layers' weight gradients wait in a FIFO of 256 x 256 output tiles and leave in grouped launches of one tile per CU
        # (engine.WgradQueue; a launch reaches back `wgrad_lag` layers), so the cotangents they read rotate through rings that deep.
        ViTEngine._init_cu_budget(self)  # n_cus, reserved_cus, cu_budget, wgrad_max_lag
        self.wgrad_tile, self.wgrad_lag = self._wgrad_group_plan()
        depth = max(2, self.wgrad_lag + 1)
        self.dbr_ring = [e(M, d, dt=bf16) for _ in range(2 * depth)]
        self.d_u_ring = [e(M, F, dt=bf16) for _ in range(depth)]
        self.dqkv_ring = [e(M, 3 * d, dt=bf16) for _ in range(depth)]
        self.dbr, self.d_u, self.dqkv = self.dbr_ring[0], self.d_u_ring[0], self.dqkv_ring[0]
        self.dsbuf = e(B, H, N, self.Np, dt=bf16)
        self.colsum_slab = e(max(1, self.L.savit_gemm_colsum_rows_cus(M, F, d, 0, self.cu_budget if self.reserved_cus else 0)), F)
        self.dcls, self.dcls_b = e(B, d), e(B, d, dt=bf16)
        self.dbr_c, self.d_hc2, self.d_oc = e(B, d, dt=bf16), e(B, d, dt=bf16), e(B, d, dt=bf16)
        self.d_uc = e(B, F, dt=bf16)
        self.dqkvc = z(Mc, 3 * d, dt=bf16)
        self.d_hc = e(Mc, d, dt=bf16)
        self.dlogits, self.d_z = z(B, self.Cp, dt=bf16), e(B, d, dt=bf16)
        ws = max(self.L.savit_layernorm_bwd_workspace_bytes(M, d), self.L.savit_th_attention_bwd_workspace_bytes(B, N, H),
                 self.L.savit_th_fused_bwd_workspace_bytes(B, N, H, cfg.head_dim), 16)
        self.ws = torch.empty(int(ws), dtype=torch.uint8, device=self.dev)
        self.labels = torch.zeros(B, dtype=torch.int32, device=self.dev)
        self.loss, self.loss_rows, self.top1, self.top5 = z(1), z(B), z(B), z(B)
        self._img_buf = e(B, cfg.img_size, cfg.img_size, 3, dt=bf16)
        self._fwd_plan = self._bwd_plan = self._cast_plan = None
        self._bwd_hooks: Dict[str, object] = {}
        self._data_parallel = False
        # round 5, as in ViTEngine: first-touch grouped weight gradients, the gradient norm's squares carried by those launches, the
        # LayerNorm / LayerScale column sums of the SA layers reduced by one launch at the end of backward (alone on the GPU only)
        self.first_touch = self.opt.first_touch
        self.defer_ln_finalize = self.opt.defer_ln_finalize
        self._gnorm_folded = False
        self._accumulate_run = False
        self.launch_timer = None  # timing.LaunchTimer (bench.py, profile_step)
        self.weights_stale = True
        self.overlap_wgrad = True if self.opt.overlap_wgrad is None else bool(self.opt.overlap_wgrad)
        self.n_side_streams = int(self.opt.side_streams)
        self.wgrad_cu_share = float(self.opt.wgrad_cu_share)
        self._side_streams = []
        self._building_serial = False
        self._bwd_plan_serial = None

    # ------------------------------------------------------------------------------------ parameters
    def param_tree(self):
        return self.layout.flax_tree(self.params)

    def grad_tree(self):
        return self.layout.flax_tree(self.grads)

    def load_params(self, tree: dict):
        src = tree["params"] if "params" in tree else tree
        _copy_tree(self.param_tree()["params"], src)
        self.weights_stale = True

    def init_params(self, seed: int = 0):
        """Reference initialisers: lecun-normal Dense, zero biases / cls / head, normal(0.02) pos-embed, LN 1/0, orthogonal
        talking-heads matrices (talking_heads.py:12), LayerScale = eps (layerscale.py:5-10)."""
        g = torch.Generator(device="cpu").manual_seed(int(seed))
        self.params.zero_()
        lay, cfg = self.layout, self.cfg

        def lecun(name, fan_in):
            std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
            t = torch.empty(lay.off[name][1], dtype=f32)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=g)
            lay.view(self.params, name).copy_(t)

        lecun("Wpe", cfg.patch_dim)
        lay.view(self.params, "pos").copy_(torch.randn(lay.off["pos"][1], generator=g) * 0.02)
        for pre, n in (("l", cfg.num_layers), ("c", cfg.num_layers_token_only)):
            for i in range(n):
                for nm in ("ln1_g", "ln2_g"):
                    lay.view(self.params, f"{pre}{i}.{nm}").fill_(1.0)
                for nm in ("ls1", "ls2"):
                    lay.view(self.params, f"{pre}{i}.{nm}").fill_(cfg.layerscale_eps)
                lecun(f"{pre}{i}.Wqkv", cfg.embed_dim)
                lecun(f"{pre}{i}.Wo", cfg.embed_dim)
                lecun(f"{pre}{i}.W1", cfg.embed_dim)
                lecun(f"{pre}{i}.W2", cfg.hidden)
                if pre == "l":
                    for nm in ("T1", "T2"):
                        q, r = torch.linalg.qr(torch.randn(cfg.num_heads, cfg.num_heads, generator=g))
                        lay.view(self.params, f"l{i}.{nm}").copy_(q * torch.sign(torch.diagonal(r)))
        lay.view(self.params, "lnf_g").fill_(1.0)
        self.weights_stale = True

    # ------------------------------------------------------------------------------------ plan helpers
    def _off_ptr(self, buf, name):
        return buf.data_ptr() + self.layout.off[name][0] * 4

    def _gemm(self, plan, label, writes=(), **kw):
        a = _lib.GemmArgs()
        for k, v in kw.items():
            setattr(a, k, v)
        if not a.rows_per_sample:
            a.rows_per_sample = 1
        a.round_bias_bf16 = self.rp
        a.cu_budget = self.cu_budget if (self.reserved_cus and self._building_bwd) else 0  # the all-reduce is resident during backward only
        plan.keep.append(a)
        plan.add(self.L.savit_gemm_bf16_tn, (ctypes.byref(a),), label, writes=writes)

    def _layer_wgrad_tiles(self, tile: int):
        d, F = self.cfg.embed_dim, self.cfg.hidden
        return [int(self.L.savit_gemm_wgrad_group_tiles(a, b, tile)) for a, b in ((F, d), (d, F), (d, d), (d, 3 * d))]  # W2, W1, Wo, Wqkv

    def _wgrad_group_plan(self):
        """(tile, layers a launch reaches back) of the SA layers' grouped weight gradients (engine.ViTEngine._wgrad_group_plan has the
        reasoning; CaiT-S24: 38 tiles per layer, a launch of 256 every ~7 layers); tile 0 (SAVIT_WGRAD_GROUP=0) = one launch per weight."""
        if not self.opt.wgrad_group:
            return 0, 0
        tile = wgrad_group_tile(self.cfg.embed_dim, self.cfg.hidden, self.opt.wgrad_tile)
        sizes = self._layer_wgrad_tiles(tile)
        q, lag = WgradQueue(self.cu_budget, self.wgrad_max_lag), 0
        for l in range(self.cfg.num_layers - 1, -1, -1):
            for t in sizes:
                q.push(None, l, t)
            while q.pending() > 0 and (q.due(l) or l == 0):
                _, _, oldest = q.take(q.cap)
                lag = max(lag, oldest - l)
        return tile, lag

    def _wgrad(self, plan, label, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, patch=(0, 0, 0, 0), side=False):
        """side=True (the big SA-layer weight gradients): no later launch consumes dW, so `_Plan.run_overlapped` issues it on a
        second stream, sized for about half the CUs (engine.ViTEngine._wgrad_splits has the measurements)."""
        splits = 0
        if side and self.overlap_wgrad and not self._building_serial:
            if self.L.savit_gemm_wgrad_auto_variant(Kin, Nout, patch[0]) == 3:  # big weights only (see ViTEngine._wgrad_splits)
                tiles = -(-Kin // 256) * -(-Nout // 256)
                splits = max(1, min(24, round(self.wgrad_cu_share * self.n_cus / tiles)))
        add_wgrad(self, plan, label, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, splits, patch, side)

    def _build_cast_plan(self):
        P, L, lay, cfg = _Plan(), self.L, self.layout, self.cfg
        d, F, C = cfg.embed_dim, cfg.hidden, cfg.num_classes
        for pre, first, stride, n in (("", "l0", lay.layer_stride, cfg.num_layers), ("c", "c0", lay.ca_stride, cfg.num_layers_token_only)):
            if n == 0:
                continue
            for nm, R, Cc in (("Wqkv", d, 3 * d), ("Wo", d, d), ("W1", d, F), ("W2", F, d)):
                P.add(L.savit_cast_transpose_bf16, (self._off_ptr(self.params, f"{first}.{nm}"), stride, n, R, Cc, self.w[pre + nm + "_n"].data_ptr(),
                                                    R * Cc, Cc, self.w[pre + nm + "_t"].data_ptr(), R * Cc, R), f"cast {pre}{nm}")
        P.add(L.savit_cast_transpose_bf16, (self._off_ptr(self.params, "Wpe"), 0, 1, cfg.patch_dim, d, None, 0, d, self.w["Wpe_t"].data_ptr(), 0,
                                            cfg.patch_dim), "cast Wpe")
        P.add(L.savit_cast_transpose_bf16, (self._off_ptr(self.params, "Wh"), 0, 1, d, C, self.w["Wh_n"].data_ptr(), 0, self.Cp,
                                            self.w["Wh_t"].data_ptr(), 0, d), "cast Wh")
        return P

    def _build_fwd_plan(self):
        P, L, cfg = _Plan(), self.L, self.cfg
        d, F, C, N, NL, NC, H, B, M, Mc = (cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.n_patches, cfg.num_layers, cfg.num_layers_token_only,
                                           cfg.num_heads, self.B, self.M, self.Mc)
        hd, Np = cfg.head_dim, self.Np
        pp = lambda n: self._off_ptr(self.params, n)  # noqa: E731
        alpha = 1.0 / math.sqrt(hd)
        x = self.x
        self._gemm(P, "patch_embed", A=self._img_buf.data_ptr(), Bt=self.w["Wpe_t"].data_ptr(), C=x[0].data_ptr(), aux=pp("pos"), M=M, N=d,
                   K=cfg.patch_dim, lda=0, ldb=cfg.patch_dim, ldc=d, ldaux=d, epilogue=_lib.EPI_PATCH, img_size=cfg.img_size, patch=cfg.patch,
                   tokens=N, token_offset=0)
        for l in range(NL):
            st = self.stats[l]
            w = lambda n, l=l: self.w[n][l].data_ptr()  # noqa: E731
            sd0, sd1 = self.sd[2 * l].data_ptr(), self.sd[2 * l + 1].data_ptr()
            P.add(L.savit_layernorm_fwd, (x[l].data_ptr(), pp(f"l{l}.ln1_g"), pp(f"l{l}.ln1_b"), self.h1[l].data_ptr(), st[0].data_ptr(),
                                          st[1].data_ptr(), M, d, d, 1e-6, self.rp), f"l{l}.ln1")
            self._gemm(P, f"l{l}.qkv", A=self.h1[l].data_ptr(), Bt=w("Wqkv_t"), C=self.qkv[l].data_ptr(), M=M, N=3 * d, K=d, lda=d, ldb=d,
                       ldc=3 * d, epilogue=_lib.EPI_BF16, alpha=alpha, alpha_cols=d)
            if self.th_fused:
                P.add(L.savit_th_fused_attention_fwd, (self.qkv[l].data_ptr(), pp(f"l{l}.T1"), pp(f"l{l}.T2"), self.o[l].data_ptr(), B, N, H, hd, 3 * d,
                                                       self.th_vt.data_ptr(), self.th_vt.numel()), f"l{l}.th_attn")
            else:
                P.add(L.savit_th_attention_fwd, (self.qkv[l].data_ptr(), pp(f"l{l}.T1"), pp(f"l{l}.T2"), self.sbuf[l].data_ptr(),
                                                 self.pbuf[l].data_ptr(), self.o[l].data_ptr(), B, N, H, hd, 3 * d, Np), f"l{l}.th_attn")
            self._gemm(P, f"l{l}.proj", A=self.o[l].data_ptr(), Bt=w("Wo_t"), C=self.xmid[l].data_ptr(), C2=self.br1[l].data_ptr(),
                       aux=x[l].data_ptr(), colscale=pp(f"l{l}.ls1"), rowscale=sd0, rows_per_sample=N, M=M, N=d, K=d, lda=d, ldb=d, ldc=d,
                       ldaux=d, epilogue=_lib.EPI_RESID)
            P.add(L.savit_layernorm_fwd, (self.xmid[l].data_ptr(), pp(f"l{l}.ln2_g"), pp(f"l{l}.ln2_b"), self.h2[l].data_ptr(), st[2].data_ptr(),
                                          st[3].data_ptr(), M, d, d, 1e-6, self.rp), f"l{l}.ln2")
            self._gemm(P, f"l{l}.fc1", A=self.h2[l].data_ptr(), Bt=w("W1_t"), C=self.u[l].data_ptr(), C2=self.a[l].data_ptr(), bias=pp(f"l{l}.b1"),
                       M=M, N=F, K=d, lda=d, ldb=d, ldc=F, epilogue=_lib.EPI_BIAS_GELU)
            self._gemm(P, f"l{l}.fc2", A=self.a[l].data_ptr(), Bt=w("W2_t"), C=x[l + 1].data_ptr(), C2=self.br2[l].data_ptr(), bias=pp(f"l{l}.b2"),
                       aux=self.xmid[l].data_ptr(), colscale=pp(f"l{l}.ls2"), rowscale=sd1, rows_per_sample=N, M=M, N=d, K=F, lda=F, ldb=F,
                       ldc=d, ldaux=d, epilogue=_lib.EPI_RESID)
        # cls token rows: cls[0][b] = cls parameter (cait.py:157-160)
        P.add(L.savit_cls_pos_rows, (pp("cls"), self.zero_d.data_ptr(), self.cls[0].data_ptr(), B, d, d), "cls_bcast")
        for c in range(NC):
            w = lambda n, c=c: self.w["c" + n][c].data_ptr()  # noqa: E731
            sd0, sd1 = self.sd[2 * (NL + c)].data_ptr(), self.sd[2 * (NL + c) + 1].data_ptr()
            cs, cx = self.cstat_c[c], self.cstat_x[c]
            # LayerNorm over concat([cls, x]) (cait.py:98-99) as two row-mapped calls into hc [B*(N+1), d]
            P.add(L.savit_layernorm_fwd_mapped, (x[NL].data_ptr(), pp(f"c{c}.ln1_g"), pp(f"c{c}.ln1_b"), self.hc[c].data_ptr(), cx[0].data_ptr(),
                                                 cx[1].data_ptr(), M, d, d, 1e-6, self.rp, N, N + 1, 1), f"c{c}.ln1x")
            P.add(L.savit_layernorm_fwd_mapped, (self.cls[c].data_ptr(), pp(f"c{c}.ln1_g"), pp(f"c{c}.ln1_b"), self.hc[c].data_ptr(),
                                                 cs[0].data_ptr(), cs[1].data_ptr(), B, d, d, 1e-6, self.rp, 1, N + 1, 0), f"c{c}.ln1c")
            qk = self.qkvc[c].data_ptr()
            # queries from the cls rows only (cait.py:14), keys / values from every row
            self._gemm(P, f"c{c}.q", A=self.hc[c].data_ptr(), Bt=w("Wqkv_t"), C=qk, M=B, N=d, K=d, lda=(N + 1) * d, ldb=d, ldc=(N + 1) * 3 * d,
                       epilogue=_lib.EPI_BF16, alpha=alpha, alpha_cols=d)
            self._gemm(P, f"c{c}.kv", A=self.hc[c].data_ptr(), Bt=self.w["cWqkv_t"][c].data_ptr() + d * d * 2, C=qk + d * 2, M=Mc, N=2 * d, K=d,
                       lda=d, ldb=d, ldc=3 * d, epilogue=_lib.EPI_BF16)
            P.add(L.savit_class_attention_fwd, (qk, (N + 1) * 3 * d, qk + d * 2, 3 * d, self.oc[c].data_ptr(), self.probs[c].data_ptr(), B, N + 1, H,
                                                hd), f"c{c}.cattn")
            self._gemm(P, f"c{c}.proj", A=self.oc[c].data_ptr(), Bt=w("Wo_t"), C=self.clsmid[c].data_ptr(), C2=self.cbr1[c].data_ptr(),
                       aux=self.cls[c].data_ptr(), colscale=pp(f"c{c}.ls1"), rowscale=sd0, rows_per_sample=1, M=B, N=d, K=d, lda=d, ldb=d, ldc=d,
                       ldaux=d, epilogue=_lib.EPI_RESID)
            P.add(L.savit_layernorm_fwd, (self.clsmid[c].data_ptr(), pp(f"c{c}.ln2_g"), pp(f"c{c}.ln2_b"), self.hc2[c].data_ptr(), cs[2].data_ptr(),
                                          cs[3].data_ptr(), B, d, d, 1e-6, self.rp), f"c{c}.ln2")
            self._gemm(P, f"c{c}.fc1", A=self.hc2[c].data_ptr(), Bt=w("W1_t"), C=self.uc[c].data_ptr(), C2=self.ac[c].data_ptr(),
                       bias=pp(f"c{c}.b1"), M=B, N=F, K=d, lda=d, ldb=d, ldc=F, epilogue=_lib.EPI_BIAS_GELU)
            self._gemm(P, f"c{c}.fc2", A=self.ac[c].data_ptr(), Bt=w("W2_t"), C=self.cls[c + 1].data_ptr(), C2=self.cbr2[c].data_ptr(),
                       bias=pp(f"c{c}.b2"), aux=self.clsmid[c].data_ptr(), colscale=pp(f"c{c}.ls2"), rowscale=sd1, rows_per_sample=1, M=B, N=d,
                       K=F, lda=F, ldb=F, ldc=d, ldaux=d, epilogue=_lib.EPI_RESID)
        # final LayerNorm: only the cls row reaches the head (cait.py:175-178)
        P.add(L.savit_layernorm_fwd, (self.cls[NC].data_ptr(), pp("lnf_g"), pp("lnf_b"), self.zcls.data_ptr(), self.fstats[0].data_ptr(),
                                      self.fstats[1].data_ptr(), B, d, d, 1e-6, self.rp), "lnf")
        self._gemm(P, "head", A=self.zcls.data_ptr(), Bt=self.w["Wh_t"].data_ptr(), C=self.logits.data_ptr(), bias=pp("bh"), M=B, N=C, K=d, lda=d,
                   ldb=d, ldc=C, epilogue=_lib.EPI_F32, round_out_bf16=self.rp)
        return P

    def _build_bwd_plan(self):
        self._building_bwd = True
        try:
            return self._record_bwd_plan()
        finally:
            self._building_bwd = False

    def _record_bwd_plan(self):
        P, L, cfg = _Plan(), self.L, self.cfg
        d, F, C, N, NL, NC, H, B, M, Mc = (cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.n_patches, cfg.num_layers, cfg.num_layers_token_only,
                                           cfg.num_heads, self.B, self.M, self.Mc)
        hd, Np = cfg.head_dim, self.Np
        pp = lambda n: self._off_ptr(self.params, n)  # noqa: E731
        gp = lambda n: self._off_ptr(self.grads, n)  # noqa: E731
        ws, wsb = self.ws.data_ptr(), self.ws.numel()
        dqs = 1.0 / math.sqrt(hd)
        # ---- head + final LayerNorm -> dcls (gradient of the cls token state)
        self._wgrad(P, "head.wgrad", self.zcls.data_ptr(), self.dlogits.data_ptr(), gp("Wh"), B, d, C, d, self.Cp, C)
        self._gemm(P, "head.dgrad", A=self.dlogits.data_ptr(), Bt=self.w["Wh_n"].data_ptr(), C=self.d_z.data_ptr(), M=B, N=d, K=self.Cp, lda=self.Cp,
                   ldb=self.Cp, ldc=d, epilogue=_lib.EPI_BF16)
        P.add(L.savit_layernorm_bwd, (self.d_z.data_ptr(), self.cls[NC].data_ptr(), pp("lnf_g"), self.fstats[0].data_ptr(), self.fstats[1].data_ptr(),
                                      None, self.dcls.data_ptr(), None, gp("lnf_g"), gp("lnf_b"), None, B, d, d, d, self.rp, ws, wsb), "lnf.bwd")
        # ---- class-attention blocks (reverse); dres accumulates the gradient w.r.t. the patch tokens x[NL] (zeroed by the caller)
        for c in range(NC - 1, -1, -1):
            w = lambda n, c=c: self.w["c" + n][c].data_ptr()  # noqa: E731
            sd0, sd1 = self.sd[2 * (NL + c)].data_ptr(), self.sd[2 * (NL + c) + 1].data_ptr()
            cs, cx = self.cstat_c[c], self.cstat_x[c]
            P.add(L.savit_layerscale_bwd, (self.dcls.data_ptr(), self.cbr2[c].data_ptr(), pp(f"c{c}.ls2"), sd1, 1, self.dbr_c.data_ptr(),
                                           gp(f"c{c}.ls2"), gp(f"c{c}.b2"), B, d, d, ws, wsb), f"c{c}.ls2.bwd")
            self._wgrad(P, f"c{c}.W2.wgrad", self.ac[c].data_ptr(), self.dbr_c.data_ptr(), gp(f"c{c}.W2"), B, F, d, F, d, d)
            self._gemm(P, f"c{c}.fc2.dgrad", A=self.dbr_c.data_ptr(), Bt=w("W2_n"), C=self.d_uc.data_ptr(), aux=self.uc[c].data_ptr(),
                       colsum=gp(f"c{c}.b1"), M=B, N=F, K=d, lda=d, ldb=d, ldc=F, ldaux=F, epilogue=_lib.EPI_DGELU)
            self._wgrad(P, f"c{c}.W1.wgrad", self.hc2[c].data_ptr(), self.d_uc.data_ptr(), gp(f"c{c}.W1"), B, d, F, d, F, F)
            self._gemm(P, f"c{c}.fc1.dgrad", A=self.d_uc.data_ptr(), Bt=w("W1_n"), C=self.d_hc2.data_ptr(), M=B, N=d, K=F, lda=F, ldb=F, ldc=d,
                       epilogue=_lib.EPI_BF16)
            P.add(L.savit_layernorm_bwd, (self.d_hc2.data_ptr(), self.clsmid[c].data_ptr(), pp(f"c{c}.ln2_g"), cs[2].data_ptr(), cs[3].data_ptr(),
                                          self.dcls.data_ptr(), self.dcls.data_ptr(), None, gp(f"c{c}.ln2_g"), gp(f"c{c}.ln2_b"), None, B, d, d, d,
                                          self.rp, ws, wsb), f"c{c}.ln2.bwd")
            P.add(L.savit_layerscale_bwd, (self.dcls.data_ptr(), self.cbr1[c].data_ptr(), pp(f"c{c}.ls1"), sd0, 1, self.dbr_c.data_ptr(),
                                           gp(f"c{c}.ls1"), None, B, d, d, ws, wsb), f"c{c}.ls1.bwd")
            self._wgrad(P, f"c{c}.Wo.wgrad", self.oc[c].data_ptr(), self.dbr_c.data_ptr(), gp(f"c{c}.Wo"), B, d, d, d, d, d)
            self._gemm(P, f"c{c}.proj.dgrad", A=self.dbr_c.data_ptr(), Bt=w("Wo_n"), C=self.d_oc.data_ptr(), M=B, N=d, K=d, lda=d, ldb=d, ldc=d,
                       epilogue=_lib.EPI_BF16)
            qk, dqk = self.qkvc[c].data_ptr(), self.dqkvc.data_ptr()
            # dqkvc: q columns are zero except at the cls rows (written here); k|v columns are written for every row
            P.add(L.savit_class_attention_bwd, (qk, (N + 1) * 3 * d, qk + d * 2, 3 * d, self.probs[c].data_ptr(), self.d_oc.data_ptr(), dqk,
                                                (N + 1) * 3 * d, dqk + d * 2, B, N + 1, H, hd, dqs), f"c{c}.cattn.bwd")
            self._wgrad(P, f"c{c}.Wqkv.wgrad", self.hc[c].data_ptr(), dqk, gp(f"c{c}.Wqkv"), Mc, d, 3 * d, d, 3 * d, 3 * d)
            self._gemm(P, f"c{c}.qkv.dgrad", A=dqk, Bt=w("Wqkv_n"), C=self.d_hc.data_ptr(), M=Mc, N=d, K=3 * d, lda=3 * d, ldb=3 * d, ldc=d,
                       epilogue=_lib.EPI_BF16)
            P.add(L.savit_layernorm_bwd_mapped, (self.d_hc.data_ptr(), self.x[NL].data_ptr(), pp(f"c{c}.ln1_g"), cx[0].data_ptr(), cx[1].data_ptr(),
                                                 self.dres.data_ptr(), self.dres.data_ptr(), None, gp(f"c{c}.ln1_g"), gp(f"c{c}.ln1_b"), None, M, d,
                                                 d, d, self.rp, N, N + 1, 1, ws, wsb), f"c{c}.ln1x.bwd")
            P.add(L.savit_layernorm_bwd_mapped, (self.d_hc.data_ptr(), self.cls[c].data_ptr(), pp(f"c{c}.ln1_g"), cs[0].data_ptr(), cs[1].data_ptr(),
                                                 self.dcls.data_ptr(), self.dcls.data_ptr(), None, gp(f"c{c}.ln1_g"), gp(f"c{c}.ln1_b"), None, B, d,
                                                 d, d, self.rp, 1, N + 1, 0, ws, wsb), f"c{c}.ln1c.bwd")
        P.add(L.savit_pos_cls_grad, (self.dcls.data_ptr(), gp("cls"), None, B, 1, d, 0), "cls.grad")
        # ---- SA layers (reverse).  Their weight gradients wait in the tile FIFO (grouped launches of one tile per CU) or, ungrouped, go
        # to the side stream; dbr / d_u / dqkv rotate through rings as deep as a launch reaches back.
        queue = WgradQueue(self.cu_budget, self.wgrad_max_lag) if self.wgrad_tile else None
        n_launch = [0]

        def wgrad_l(label, layer, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw):
            if queue is None:
                return self._wgrad(P, label, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, side=True)
            queue.push((X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw), layer, int(L.savit_gemm_wgrad_group_tiles(Kin, Nout, self.wgrad_tile)))

        def flush_group(layer: int, last: bool):
            # after a layer's last input-gradient GEMM, before anything overwrites the oldest ring slots; the DDP trigger of an EARLIER
            # layer ('l{j}.ln1.bwd') whose last tile is in a launch fires behind that launch
            while queue is not None and queue.pending() > 0 and (queue.due(layer) or last):
                entries, done, oldest = queue.take(queue.cap)
                assert oldest - layer <= self.wgrad_lag, "weight-gradient queue reaches back further than the cotangent rings"
                add_wgrad_group(self, P, f"wgrad.group.{n_launch[0]}.l{oldest}-l{layer}", entries, self.wgrad_tile,
                                [f"l{j}.ln1.bwd" for j in done if j != layer], side=False)
                n_launch[0] += 1

        ring, ri = [t.data_ptr() for t in self.dbr_ring], 0
        fuse_ls = d > 64  # savit_layernorm_bwd_ls serves wide rows (every head_dim 48 / 64 geometry)
        # alone on the GPU the fused LayerNorm + LayerScale backward launches of the SA layers leave their column-sum slabs (four
        # families) to ONE finalize launch at the end of backward (47 finalize launches + kernel boundaries per CaiT-S24 step)
        defer = self.defer_ln_finalize and not self._data_parallel and fuse_ls
        jobs: List[tuple] = []

        def ln_bwd_ls(label, head7, outs2, mid, tail, outs_ls, extra, writes):
            """savit_layernorm_bwd_ls; head7 = dy .. dx, outs2 = (dgamma, dbeta), mid = rows .. round, tail = branch .. dbranch,
            outs_ls = (d_layerscale, dbias), extra = (slab, rows, n, out) or (None, 0, 0, None)"""
            if defer:
                wb = self._ln_ws_slot(len(jobs), mid[0], d)
                P.add(L.savit_layernorm_bwd_ls, head7 + (None, None) + mid + tail + (None, None, wb.data_ptr(), wb.numel(), None, 0, 0, None), label,
                      writes=writes)
                jobs.append((wb.data_ptr(), int(L.savit_layernorm_bwd_grid(mid[0])), d, 4, outs2 + outs_ls, extra if extra[0] is not None else None))
            else:
                P.add(L.savit_layernorm_bwd_ls, head7 + outs2 + mid + tail + outs_ls + (ws, wsb) + extra, label, writes=writes)
        for l in range(NL - 1, -1, -1):
            st = self.stats[l]
            w = lambda n, l=l: self.w[n][l].data_ptr()  # noqa: E731
            sd0, sd1 = self.sd[2 * l].data_ptr(), self.sd[2 * l + 1].data_ptr()
            d_u, dqkv = self.d_u_ring[l % len(self.d_u_ring)].data_ptr(), self.dqkv_ring[l % len(self.dqkv_ring)].data_ptr()
            if l == NL - 1 or not fuse_ls:  # below the first layer processed there is no LayerNorm backward to ride on
                ri = (ri + 1) % len(ring)
                P.add(L.savit_layerscale_bwd, (self.dres.data_ptr(), self.br2[l].data_ptr(), pp(f"l{l}.ls2"), sd1, N, ring[ri], gp(f"l{l}.ls2"),
                                               gp(f"l{l}.b2"), M, d, d, ws, wsb), f"l{l}.ls2.bwd", writes=(ring[ri],))
            wgrad_l(f"l{l}.W2.wgrad", l, self.a[l].data_ptr(), ring[ri], gp(f"l{l}.W2"), M, F, d, F, d, d)
            # bias gradient: per-row-tile partial sums to a slab + finalize, as in the ViT engine (atomic column sums cost this launch
            # 160 instead of 117 us at CaiT-S24: 50 k rows adding into 1536 addresses)
            cslab = self._colsum_slab_for(l) if defer else self.colsum_slab  # (deferred: the slab lives until the end of backward)
            self._gemm(P, f"l{l}.fc2.dgrad", writes=(d_u,), A=ring[ri], Bt=w("W2_n"), C=d_u, aux=self.u[l].data_ptr(),
                       colsum=cslab.data_ptr(), colsum_rows=cslab.shape[0], M=M, N=F, K=d, lda=d, ldb=d, ldc=F,
                       ldaux=F, epilogue=_lib.EPI_DGELU)
            if not fuse_ls:  # (fused: the slab's column sums ride along with this layer's ln2.bwd finalize)
                P.add(L.savit_colsum_finalize, (self.colsum_slab.data_ptr(), self.colsum_slab.shape[0], F, gp(f"l{l}.b1"), 1), f"l{l}.b1.grad")
            wgrad_l(f"l{l}.W1.wgrad", l, self.h2[l].data_ptr(), d_u, gp(f"l{l}.W1"), M, d, F, d, F, F)
            self._gemm(P, f"l{l}.fc1.dgrad", A=d_u, Bt=w("W1_n"), C=self.d_h.data_ptr(), M=M, N=d, K=F, lda=F, ldb=F, ldc=d,
                       epilogue=_lib.EPI_BF16)
            # LayerNorm backward + the LayerScale backward of the sub-block before it in ONE pass over the residual gradient
            # (savit_layernorm_bwd_ls; two launches re-read it: 30 + 5 us per sub-block at CaiT-S24)
            ri = (ri + 1) % len(ring)
            if not fuse_ls:
                P.add(L.savit_layernorm_bwd, (self.d_h.data_ptr(), self.xmid[l].data_ptr(), pp(f"l{l}.ln2_g"), st[2].data_ptr(), st[3].data_ptr(),
                                              self.dres.data_ptr(), self.dres.data_ptr(), None, gp(f"l{l}.ln2_g"), gp(f"l{l}.ln2_b"), None, M, d, d,
                                              d, self.rp, ws, wsb), f"l{l}.ln2.bwd")
                P.add(L.savit_layerscale_bwd, (self.dres.data_ptr(), self.br1[l].data_ptr(), pp(f"l{l}.ls1"), sd0, N, ring[ri], gp(f"l{l}.ls1"),
                                               None, M, d, d, ws, wsb), f"l{l}.ls1.bwd", writes=(ring[ri],))
            else:
                ln_bwd_ls(f"l{l}.ln2.bwd", (self.d_h.data_ptr(), self.xmid[l].data_ptr(), pp(f"l{l}.ln2_g"), st[2].data_ptr(), st[3].data_ptr(),
                                            self.dres.data_ptr(), self.dres.data_ptr()), (gp(f"l{l}.ln2_g"), gp(f"l{l}.ln2_b")), (M, d, d, d, self.rp),
                          (self.br1[l].data_ptr(), pp(f"l{l}.ls1"), sd0, N, ring[ri]), (gp(f"l{l}.ls1"), None),
                          (cslab.data_ptr(), cslab.shape[0], F, gp(f"l{l}.b1")), (ring[ri],))
            wgrad_l(f"l{l}.Wo.wgrad", l, self.o[l].data_ptr(), ring[ri], gp(f"l{l}.Wo"), M, d, d, d, d, d)
            self._gemm(P, f"l{l}.proj.dgrad", A=ring[ri], Bt=w("Wo_n"), C=self.d_o.data_ptr(), M=M, N=d, K=d, lda=d, ldb=d, ldc=d,
                       epilogue=_lib.EPI_BF16)
            if self.th_fused:
                P.add(L.savit_th_fused_attention_bwd, (self.qkv[l].data_ptr(), pp(f"l{l}.T1"), pp(f"l{l}.T2"), self.d_o.data_ptr(),
                                                       self.th_pbuf.data_ptr(), self.dsbuf.data_ptr(), dqkv, gp(f"l{l}.T1"), gp(f"l{l}.T2"), B, N, H,
                                                       hd, 3 * d, Np, dqs, ws, wsb), f"l{l}.th_attn.bwd", writes=(dqkv,))
            else:
                P.add(L.savit_th_attention_bwd, (self.qkv[l].data_ptr(), pp(f"l{l}.T1"), pp(f"l{l}.T2"), self.sbuf[l].data_ptr(),
                                                 self.pbuf[l].data_ptr(), self.d_o.data_ptr(), self.dsbuf.data_ptr(), dqkv, gp(f"l{l}.T1"),
                                                 gp(f"l{l}.T2"), B, N, H, hd, 3 * d, Np, dqs, ws, wsb), f"l{l}.th_attn.bwd", writes=(dqkv,))
            wgrad_l(f"l{l}.Wqkv.wgrad", l, self.h1[l].data_ptr(), dqkv, gp(f"l{l}.Wqkv"), M, d, 3 * d, d, 3 * d, 3 * d)
            self._gemm(P, f"l{l}.qkv.dgrad", A=dqkv, Bt=w("Wqkv_n"), C=self.d_h.data_ptr(), M=M, N=d, K=3 * d, lda=3 * d, ldb=3 * d,
                       ldc=d, epilogue=_lib.EPI_BF16)
            flush_group(l, last=(l == 0))
            if l > 0 and fuse_ls:  # ... and layer l-1's second sub-block behind this layer's first LayerNorm
                ri = (ri + 1) % len(ring)
                ln_bwd_ls(f"l{l}.ln1.bwd", (self.d_h.data_ptr(), self.x[l].data_ptr(), pp(f"l{l}.ln1_g"), st[0].data_ptr(), st[1].data_ptr(),
                                            self.dres.data_ptr(), self.dres.data_ptr()), (gp(f"l{l}.ln1_g"), gp(f"l{l}.ln1_b")), (M, d, d, d, self.rp),
                          (self.br2[l - 1].data_ptr(), pp(f"l{l - 1}.ls2"), self.sd[2 * l - 1].data_ptr(), N, ring[ri]),
                          (gp(f"l{l - 1}.ls2"), gp(f"l{l - 1}.b2")), (None, 0, 0, None), (ring[ri],))
            else:  # the bf16 copy feeds the patch-embed weight gradient
                P.add(L.savit_layernorm_bwd, (self.d_h.data_ptr(), self.x[l].data_ptr(), pp(f"l{l}.ln1_g"), st[0].data_ptr(), st[1].data_ptr(),
                                              self.dres.data_ptr(), self.dres.data_ptr(), self.dres_b.data_ptr(), gp(f"l{l}.ln1_g"),
                                              gp(f"l{l}.ln1_b"), None, M, d, d, d, self.rp, ws, wsb), f"l{l}.ln1.bwd")
        P.add(L.savit_pos_cls_grad, (self.dres.data_ptr(), gp("pos"), None, B, N, d, 0), "pos.grad")
        self._wgrad(P, "Wpe.wgrad", self._img_buf.data_ptr(), self.dres_b.data_ptr(), gp("Wpe"), M, cfg.patch_dim, d, 0, d, d,
                    patch=(cfg.patch, cfg.img_size, N, 0))
        if jobs:
            arr = (_lib.ColsumJob * len(jobs))()
            for q, (partial, nblk, dd, nf, outs, extra) in zip(arr, jobs):
                q.partial, q.nblk, q.d, q.nf = partial, nblk, dd, nf
                for i, o in enumerate(outs):
                    q.out[i] = o
                if extra is not None:
                    q.extra_slab, q.extra_rows, q.extra_n, q.extra_out = extra
            P.keep.append(arr)
            P.add(L.savit_layernorm_bwd_finalize_jobs, (arr, len(jobs)), "ln.bwd.finalize")
        finalize_wgrad_ws(self, P)
        finalize_first_touch(self, P)
        return P

    # ------------------------------------------------------------------------------------ execution
    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def refresh_weights(self):
        if self._cast_plan is None:
            self._cast_plan = self._build_cast_plan()
        self._cast_plan.run(self._stream(), self.launch_timer)
        self.weights_stale = False

    def set_images(self, images: torch.Tensor):
        S = self.cfg.img_size
        if not images.is_cuda:
            raise ValueError("images must be on the GPU")
        if tuple(images.shape) == (self.B, S, S, 3) and images.dtype == bf16:
            self._img_buf.copy_(images)
        elif tuple(images.shape) == (self.B, S, S, 3) and images.dtype == f32:
            _lib.check(self.L.savit_cast_bf16(images.contiguous().data_ptr(), self._img_buf.data_ptr(), images.numel(), self._stream()), "savit_cast_bf16")
        elif tuple(images.shape) == (S, S, 3, self.B) and images.dtype == f32:
            _lib.check(self.L.savit_hwcn_to_nhwc_bf16(images.contiguous().data_ptr(), self._img_buf.data_ptr(), S, S, 3, self.B, self._stream()),
                       "savit_hwcn_to_nhwc_bf16")
        else:
            raise ValueError(f"images shape {tuple(images.shape)} / dtype {images.dtype} not accepted")

    def set_stochastic_depth(self, is_training: bool, keep_masks: Optional[torch.Tensor] = None, seed: Optional[int] = None):
        """rowscale = mask / keep_prob per (layer, branch, sample) (stochastic_depth.py:16-27); identity in eval or at rate 0.
        keep_masks [(L+Lc), 2, B] of 0/1 overrides the generator (tests)."""
        rate = self.cfg.stoch_depth_rate
        if not is_training or rate == 0.0:
            self.sd.fill_(1.0)
            return
        keep = 1.0 - rate
        if keep_masks is not None:
            self.sd.copy_(keep_masks.to(device=self.dev, dtype=f32).reshape(-1, self.B) / keep)
            return
        if seed is not None:
            self.gen.manual_seed(int(seed))
        u = torch.rand(self.sd.shape, device=self.dev, generator=self.gen)
        self.sd.copy_(torch.floor(keep + u) / keep)

    def forward(self, images: Optional[torch.Tensor] = None, is_training: bool = False, keep_masks=None,
                sd_seed: Optional[int] = None) -> torch.Tensor:
        """sd_seed: seed of THIS step's stochastic-depth masks (see stochastic_depth_seed); None continues the engine's stream."""
        if images is not None:
            self.set_images(images)
        self.set_stochastic_depth(is_training, keep_masks, seed=sd_seed)
        if self.weights_stale:
            self.refresh_weights()
        if self._fwd_plan is None:
            self._fwd_plan = self._build_fwd_plan()
        self._fwd_plan.run(self._stream(), self.launch_timer)
        return self.logits

    # (the plan / gradient-buffer plumbing of ViTEngine, which this class does not derive from)
    bwd_hooks = ViTEngine.bwd_hooks
    _fold_sumsq_ok = ViTEngine._fold_sumsq_ok
    _current_bwd_plan = ViTEngine._current_bwd_plan
    _zero_grads_for_backward = ViTEngine._zero_grads_for_backward
    _grad_sumsq = ViTEngine._grad_sumsq
    _ln_ws_slot = ViTEngine._ln_ws_slot
    _colsum_slab_for = ViTEngine._colsum_slab_for

    def loss_backward(self, labels, label_smoothing: float = 0.1, mix_labels=None, ratio=None, zero_grads: bool = True):
        s = self._stream()
        self.labels.copy_(labels.to(torch.int32))
        self._zero_grads_for_backward(zero_grads)
        self._zero("zero.loss", self.loss)
        ml = mr = None
        if mix_labels is not None:
            self._mix_labels = mix_labels.to(device=self.dev, dtype=torch.int32).contiguous()
            self._mix_ratio = ratio.to(device=self.dev, dtype=f32).contiguous()
            ml, mr = self._mix_labels.data_ptr(), self._mix_ratio.data_ptr()
        timed_call(self.launch_timer, "xent", self.L.savit_softmax_xent, self.logits.data_ptr(), self.cfg.num_classes, self.labels.data_ptr(), ml, mr,
                   float(label_smoothing), 1.0 / self.B, self.loss_rows.data_ptr(), self.loss.data_ptr(), self.dlogits.data_ptr(), self.Cp,
                   self._off_ptr(self.grads, "bh"), self.top1.data_ptr(), self.top5.data_ptr(), self.B, self.cfg.num_classes, s)
        self._zero("zero.dres", self.dres)
        ViTEngine._run_bwd(self)
        return self.loss

    def _zero(self, label: str, t: torch.Tensor):
        timed_call(self.launch_timer, label, self.L.savit_zero_bytes, t.data_ptr(), t.numel() * t.element_size(), self._stream())

    def _serial_bwd_plan(self):
        if self._bwd_plan_serial is None:
            self._building_serial = True
            try:
                self._bwd_plan_serial = self._build_bwd_plan()
            finally:
                self._building_serial = False
        return self._bwd_plan_serial

    def optimizer_step(self, lr: float, weight_decay: float = 0.0, max_norm: float = 0.0, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8,
                       grad_scale: float = 1.0):
        if self.adam_m is None:
            self.adam_m, self.adam_v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        s = self._stream()
        self.step_count += 1
        ss = None
        tm = self.launch_timer
        if max_norm and max_norm > 0:
            ss = self._grad_sumsq(tm, s)
        self._gnorm_folded = False
        timed_call(tm, "adamw", self.L.savit_adamw_step, self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(),
                   self.params.numel(), float(lr), float(b1), float(b2), float(eps), float(weight_decay), self.step_count, ss,
                   float(max_norm or 0.0), float(grad_scale), s)
        self.refresh_weights()

    def profile_step(self, labels, label_smoothing: float = 0.1, reps: int = 3, is_training: bool = False):
        """Forward + loss + backward with every launch bracketed (timing.instrumented_steps) -> {launch label: ms, min over reps}."""
        from .timing import instrumented_steps

        def one():
            self.forward(is_training=is_training)
            self.loss_backward(labels, label_smoothing)

        return instrumented_steps(self, one, reps=reps)["labels"]
