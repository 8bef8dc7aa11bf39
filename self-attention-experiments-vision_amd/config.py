"""Model configurations of the hot path (reference: /root/reference/models/create_model.py:10-37 ViT
branches, :79-168 CaiT branches) plus the two DeiT sizes BASELINE.json names that the reference lacks."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict


@dataclass(frozen=True)
class ModelConfig:
    kind: str  # 'vit' | 'cait' | 'mixer' | 'tnt'
    num_layers: int
    num_heads: int
    embed_dim: int
    patch: int
    num_classes: int = 1000
    img_size: int = 224
    expand_ratio: float = 4.0
    num_layers_token_only: int = 0  # CaiT class-attention layers (cait.py:162)
    stoch_depth_rate: float = 0.0
    layerscale_eps: float = 0.0
    tokens_expand_ratio: float = 0.5  # MLP-Mixer token-mixing FFBlock (mlp_mixer.py:39)
    # TNT (tnt.py:139-147): num_heads / embed_dim describe the OUTER transformer
    inner_num_heads: int = 0
    inner_embed_dim: int = 0
    transformed_patch: int = 4

    @property
    def n_patches(self) -> int:
        return (self.img_size // self.patch) ** 2

    @property
    def n_pixels(self) -> int:
        """TNT: pixel tokens per patch = (patch / transformed_patch)^2 (tnt.py:24-29)."""
        return (self.patch // self.transformed_patch) ** 2

    @property
    def tokens_hidden(self) -> int:
        """hidden width of the token-mixing FFBlock: ff.py:24 with in_ch = number of patches."""
        return max(1, int(self.tokens_expand_ratio * self.n_patches))

    @property
    def seq_len(self) -> int:
        """tokens in the self-attention encoder: patches (+ cls for ViT: vit.py:85)."""
        return self.n_patches + (1 if self.kind in ("vit", "tnt") else 0)

    @property
    def hidden(self) -> int:
        """FFBlock hidden width, ff.py:24."""
        return max(1, int(self.expand_ratio * self.embed_dim))

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def patch_dim(self) -> int:
        return self.patch * self.patch * 3


def _vit(L, H, d, p):
    return dict(kind="vit", num_layers=L, num_heads=H, embed_dim=d, patch=p)


def _cait(L, H, d, sd, eps):
    return dict(kind="cait", num_layers=L, num_heads=H, embed_dim=d, patch=16, num_layers_token_only=2,
                stoch_depth_rate=sd, layerscale_eps=eps)


def _mixer(L, d, p):
    return dict(kind="mixer", num_layers=L, num_heads=1, embed_dim=d, patch=p)


def _tnt(L, Hi, Ho, di, do):
    return dict(kind="tnt", num_layers=L, num_heads=Ho, embed_dim=do, patch=16, inner_num_heads=Hi, inner_embed_dim=di)


MODEL_ZOO: Dict[str, dict] = {
    "tnt_s_patch16": _tnt(12, 4, 10, 40, 640),  # create_model.py:50-56
    "tnt_b_patch16": _tnt(12, 4, 6, 24, 384),   # :57-63
    # create_model.py:184-213; the branch at :199-203 repeats 'mixer_s_patch32' (unreachable) with Mixer-B/16 arguments and is
    # registered under the name it was meant to have
    "mixer_s_patch32": _mixer(8, 512, 32),
    "mixer_s_patch16": _mixer(8, 512, 16),
    "mixer_b_patch32": _mixer(12, 768, 32),
    "mixer_b_patch16": _mixer(12, 768, 16),
    "mixer_l_patch32": _mixer(24, 1024, 32),
    "mixer_l_patch16": _mixer(32, 1024, 16),
    "vit_b_patch32": _vit(12, 12, 768, 32),   # create_model.py:10-16
    "vit_b_patch16": _vit(12, 12, 768, 16),   # :17-23  (== DeiT-B/16, BASELINE config 3)
    "vit_l_patch32": _vit(24, 16, 1024, 32),  # :24-30
    "vit_l_patch16": _vit(24, 16, 1024, 16),  # :31-37  (BASELINE config 5 at img_size=384)
    "vit_ti_patch16": _vit(12, 3, 192, 16),   # BASELINE config 1 (absent from the reference)
    "vit_s_patch16": _vit(12, 6, 384, 16),    # BASELINE config 2, DeiT-S (absent from the reference)
    "cait_xxs_24": _cait(24, 4, 192, 0.05, 1e-5),
    "cait_xxs_36": _cait(36, 4, 192, 0.1, 1e-6),
    "cait_xs_24": _cait(24, 6, 288, 0.05, 1e-5),
    "cait_xs_36": _cait(36, 6, 288, 0.1, 1e-6),
    "cait_s_24": _cait(24, 8, 384, 0.1, 1e-6),
    "cait_s_36": _cait(36, 8, 384, 0.2, 1e-6),
    "cait_s_48": _cait(48, 8, 384, 0.3, 1e-6),
    "cait_m_24": _cait(24, 16, 768, 0.2, 1e-5),
    "cait_m_36": _cait(36, 16, 768, 0.3, 1e-6),
    "cait_m_48": _cait(48, 16, 768, 0.4, 1e-6),
}


def get_config(model_name: str, num_classes: int = 1000, img_size: int = 224) -> ModelConfig:
    if model_name not in MODEL_ZOO:
        raise RuntimeError("Model not found.")  # create_model.py:214-215
    return ModelConfig(num_classes=num_classes, img_size=img_size, **MODEL_ZOO[model_name])


def train_flops_per_image(cfg: ModelConfig) -> float:
    """Algorithmic FLOPs of one train step per image (no recompute counted)."""
    d, C, n = cfg.embed_dim, cfg.num_classes, cfg.n_patches
    pe = 2.0 * n * cfg.patch_dim * d
    if cfg.kind == "vit":
        N = n + 1
        layer = 24.0 * N * d * d + 4.0 * N * N * d
        return 3.0 * (cfg.num_layers * layer + 2.0 * d * C) + 2.0 * pe
    if cfg.kind == "tnt":
        di, npx, N = cfg.inner_embed_dim, cfg.n_pixels, n + 1
        inner = n * (24.0 * npx * di * di + 4.0 * npx * npx * di)
        i2o = 2.0 * n * (npx * di) * d
        outer = 24.0 * N * d * d + 4.0 * N * N * d
        pix = 2.0 * n * npx * (3 * cfg.transformed_patch ** 2) * di
        return 3.0 * (cfg.num_layers * (inner + i2o + outer) + 2.0 * d * C) + 2.0 * (pe + pix)
    if cfg.kind == "mixer":  # token FF 2 x (2 d n Ft) + channel FF 2 x (2 n d F) per layer
        layer = 4.0 * d * n * cfg.tokens_hidden + 4.0 * n * d * cfg.hidden
        return 3.0 * (cfg.num_layers * layer + 2.0 * d * C) + 2.0 * pe
    H = cfg.num_heads
    sa = 24.0 * n * d * d + 4.0 * n * n * d + 4.0 * H * H * n * n
    ca = 20.0 * d * d + 4.0 * (n + 1) * d * d + 4.0 * (n + 1) * d
    return 3.0 * (cfg.num_layers * sa + cfg.num_layers_token_only * ca + 2.0 * d * C) + 2.0 * pe


def cls_only_saved_flops_per_image(cfg: ModelConfig, forward_too: bool) -> float:
    """FLOPs of `train_flops_per_image` that the ViT engines do NOT execute because only the cls row of the last encoder layer's output
    is ever read (vit.py:57,95; engine.cls_only_last / cls_fwd, round 5): the last layer's output projection and MLP - and, with
    forward_too, the attention of its non-cls queries - on (N - 1) of N rows.  Priced exactly as `train_flops_per_image` prices them
    (backward = 2 x forward for every product, the attention included: ADVICE r5 - round 5 used 2.5 x for the attention backward, which is
    what the kernel recomputes, not what the count it is subtracted from contains)."""
    if cfg.kind != "vit":
        return 0.0
    d, N = cfg.embed_dim, cfg.n_patches + 1
    rows = N - 1
    dense = 2.0 * rows * d * d + 4.0 * rows * d * cfg.hidden  # proj + fc1 + fc2 on the rows left out
    attn_fwd = 4.0 * rows * N * d                              # QK^T and PV of the queries left out
    # backward (input-gradient + weight-gradient products of proj / fc1 / fc2: 2 x) always; the forward and the attention of the non-cls
    # queries (forward + 2 x backward) only when the cls-query kernels replace the dense ones
    return 2.0 * dense + ((dense + 3.0 * attn_fwd) if forward_too else 0.0)


def executed_flops_per_image(cfg: ModelConfig, cls_only_last: bool = False, cls_fwd: bool = False) -> float:
    """The algorithmic FLOPs of the step the engines actually launch: SURVEY 8d's count minus what the cls-row plan leaves out."""
    return train_flops_per_image(cfg) - (cls_only_saved_flops_per_image(cfg, cls_fwd) if cls_only_last else 0.0)
