"""fp32 arithmetic mode of the ViT forward path (forward + loss).

The reference's `create_model(model_name, num_classes=1000, dtype=jnp.float32)` defaults to float32
(/root/reference/models/create_model.py:6-8) and BASELINE config 1 is ViT-Tiny/16 in fp32: with dtype=float32 every Dense,
LayerNorm, softmax and GELU of models/vit.py:73-99 computes and returns fp32.  This engine runs that graph on the GPU with the
fp32 kernels of csrc/fp32_path.hip (exact fp32-input MFMA products on the fp32 master weights in place - no bf16 anywhere).
It shares the parameter layout (ParamLayout: flat fp32 buffer, Flax-shaped views) with the bf16 training engine, so a tree moves
between the two unchanged.  Training (backward, optimizer) is the bf16 MFMA engine's job: `loss_backward` raises here.
Limits: the ViT family, seq_len <= 256, head_dim <= 64 (every 224x224 create_model ViT)."""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import lib as _lib
from .config import ModelConfig
from .engine import ParamLayout, _Plan, _copy_tree

f32 = torch.float32


class ViTEngineF32:
    def __init__(self, cfg: ModelConfig, batch: int, device: str = "cuda"):
        if cfg.kind != "vit":
            raise NotImplementedError("fp32 arithmetic is implemented for the ViT family; the other families compute in bf16 "
                                      "(pass dtype=torch.bfloat16 to create_model)")
        if cfg.seq_len > 256 or cfg.head_dim > 64:
            raise NotImplementedError("fp32 attention keeps one head's K and V in LDS: seq_len <= 256, head_dim <= 64")
        if not torch.cuda.is_available():
            raise RuntimeError("ViTEngineF32 needs a GPU: there is no CPU path")
        self.L = _lib.load()
        self.cfg, self.B, self.dev = cfg, int(batch), torch.device(device)
        self.layout = ParamLayout(cfg)
        d, F, C, N, NL = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers
        self.M = M = self.B * N
        e = lambda *s: torch.empty(*s, dtype=f32, device=self.dev)  # noqa: E731
        self.params = torch.zeros(self.layout.total, dtype=f32, device=self.dev)
        self.grads = None
        self.w = {}
        self.adam_m = self.adam_v = None
        self.weights_stale = False
        self.images = e(self.B, cfg.img_size, cfg.img_size, 3)
        self.patches = e(self.B * cfg.n_patches, cfg.patch_dim)
        self.tok = e(self.B * cfg.n_patches, d)
        self.x = e(M, d)
        self.xmid = e(M, d)
        self.h = e(M, d)
        self.qkv = e(M, 3 * d)
        self.o = e(M, d)
        self.a = e(M, F)
        self.zcls = e(self.B, d)
        self.logits = e(self.B, C)
        self.labels = torch.zeros(self.B, dtype=torch.int32, device=self.dev)
        self.loss = torch.zeros(1, dtype=f32, device=self.dev)
        self.loss_rows, self.top1, self.top5 = e(self.B), e(self.B), e(self.B)
        self._plan: Optional[_Plan] = None

    # ---- parameters (same tree as the bf16 engine)
    def param_tree(self) -> dict:
        return self.layout.flax_tree(self.params)

    def load_params(self, tree: dict):
        _copy_tree(self.param_tree()["params"], tree["params"] if "params" in tree else tree)

    def init_params(self, seed: int = 0):
        from .engine import ViTEngine

        ViTEngine.init_params(self, seed)  # reference initialisers; touches only .params / .layout / .cfg / .weights_stale

    def _off(self, name: str) -> int:
        return self.params.data_ptr() + self.layout.off[name][0] * 4

    def _build_plan(self) -> _Plan:
        P, L, cfg = _Plan(), self.L, self.cfg
        d, F, C, N, NL, H, B, M = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers, cfg.num_heads, self.B, self.M
        pp = self._off
        x, xm, h = self.x.data_ptr(), self.xmid.data_ptr(), self.h.data_ptr()

        def gemm(label, A, W, Cc, M_, N_, K_, lda, ldw, ldc, bias=None, aux=None, ldaux=0, alpha=1.0, alpha_cols=0, gelu=0):
            P.add(L.savit_gemm_f32, (A, W, Cc, bias, aux, M_, N_, K_, lda, ldw, ldc, ldaux, alpha, alpha_cols, gelu), label)

        P.add(L.savit_patchify_f32, (self.images.data_ptr(), self.patches.data_ptr(), B, cfg.img_size, cfg.patch), "patchify")
        gemm("patch_embed", self.patches.data_ptr(), pp("Wpe"), self.tok.data_ptr(), B * cfg.n_patches, d, cfg.patch_dim, cfg.patch_dim, d, d)
        P.add(L.savit_assemble_tokens_f32, (self.tok.data_ptr(), pp("cls"), pp("pos"), x, B, N, d), "tokens")
        for l in range(NL):
            P.add(L.savit_layernorm_fwd_f32, (x, pp(f"l{l}.ln1_g"), pp(f"l{l}.ln1_b"), h, M, d, d, d, 1e-6), f"l{l}.ln1")
            gemm(f"l{l}.qkv", h, pp(f"l{l}.Wqkv"), self.qkv.data_ptr(), M, 3 * d, d, d, 3 * d, 3 * d, alpha=1.0 / math.sqrt(cfg.head_dim),
                 alpha_cols=d)
            P.add(L.savit_attention_fwd_f32, (self.qkv.data_ptr(), self.o.data_ptr(), B, N, H, cfg.head_dim, 3 * d), f"l{l}.attn")
            gemm(f"l{l}.proj", self.o.data_ptr(), pp(f"l{l}.Wo"), xm, M, d, d, d, d, d, aux=x, ldaux=d)
            P.add(L.savit_layernorm_fwd_f32, (xm, pp(f"l{l}.ln2_g"), pp(f"l{l}.ln2_b"), h, M, d, d, d, 1e-6), f"l{l}.ln2")
            gemm(f"l{l}.fc1", h, pp(f"l{l}.W1"), self.a.data_ptr(), M, F, d, d, F, F, bias=pp(f"l{l}.b1"), gelu=1)
            gemm(f"l{l}.fc2", self.a.data_ptr(), pp(f"l{l}.W2"), x, M, d, F, F, d, d, bias=pp(f"l{l}.b2"), aux=xm, ldaux=d)
        P.add(L.savit_layernorm_fwd_f32, (x, pp("lnf_g"), pp("lnf_b"), self.zcls.data_ptr(), B, d, N * d, d, 1e-6), "lnf")  # cls rows only
        gemm("head", self.zcls.data_ptr(), pp("Wh"), self.logits.data_ptr(), B, C, d, d, C, C, bias=pp("bh"))
        return P

    def set_images(self, images: torch.Tensor):
        S = self.cfg.img_size
        if not images.is_cuda or tuple(images.shape) != (self.B, S, S, 3):
            raise ValueError(f"images must be a GPU tensor [B={self.B},{S},{S},3] (NHWC)")
        self.images.copy_(images.to(f32))

    def forward(self, images: Optional[torch.Tensor] = None) -> torch.Tensor:
        if images is not None:
            self.set_images(images)
        if self._plan is None:
            self._plan = self._build_plan()
        self._plan.run(torch.cuda.current_stream().cuda_stream)
        return self.logits

    def loss_fn(self, labels: torch.Tensor, label_smoothing: float = 0.1) -> torch.Tensor:
        """train.py:83-90 on the fp32 logits (one-hot, label smoothing, softmax cross-entropy, mean); also fills top1 / top5."""
        self.labels.copy_(labels.to(torch.int32))
        self.loss.zero_()
        _lib.check(self.L.savit_softmax_xent(self.logits.data_ptr(), self.cfg.num_classes, self.labels.data_ptr(), None, None,
                                             float(label_smoothing), 1.0 / self.B, self.loss_rows.data_ptr(), self.loss.data_ptr(),
                                             None, 0, None, self.top1.data_ptr(), self.top5.data_ptr(),
                                             self.B, self.cfg.num_classes, torch.cuda.current_stream().cuda_stream), "savit_softmax_xent")
        return self.loss

    def loss_backward(self, *a, **k):
        raise NotImplementedError("fp32 arithmetic covers forward + loss; training runs on the bf16 MFMA engine "
                                  "(create_model(..., dtype=torch.bfloat16))")

    optimizer_step = backward_from_dlogits = loss_backward
