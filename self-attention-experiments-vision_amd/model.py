"""Drop-in boundary: the reference's model factory and module protocol for the hot path.

Mirrors /root/reference/models/create_model.py:6-8 (`create_model(model_name, num_classes=1000, dtype=...)`,
RuntimeError('Model not found.') on unknown names) and the Flax calling convention its callers use
(train.py:29-37,82,115; models/vit_test.py:23-26):

    model  = create_model('vit_b_patch16', num_classes=1000, dtype=torch.bfloat16)   # dtype defaults to float32, as in the reference
    params = model.init(seed, torch.ones(1, 224, 224, 3), is_training=False)        # {'params': Flax-shaped tree}
    logits = model.apply(params, images_NHWC, is_training=True)                      # [B, num_classes]
    logits, params = model.init_with_output(seed, x, is_training=True)               # as in the reference tests
    logits = model(images_NHWC, is_training)                                         # module(inputs, is_training)

Every tensor lives on the GPU; the arithmetic is the HIP engine (engine.py).  There is no CPU execution path.

dtype (create_model.py:6-8: `dtype=jnp.float32` by default; train.py passes bfloat16):
  torch.bfloat16  the training path: bf16 MFMA kernels with fp32 residual stream / statistics / parameters (all four families);
  torch.float32   the reference default: every op in fp32 (engine_f32.py, exact fp32-input MFMA) - forward, loss and train step, all four families.
                  Training entry points raise on it, and the other families raise at construction: they compute in bf16 only.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from .config import ModelConfig, get_config
from .engine import ViTEngine, _copy_tree


def _clone_tree(t):
    if isinstance(t, dict):
        return {k: _clone_tree(v) for k, v in t.items()}
    return t.detach().clone()


class ViT:
    """models/vit.py:61-99 behind the HIP engine.  Stateless w.r.t. parameters in `apply` (the caller owns the
    tree, as with Flax); `init` / `bind` attach a tree for the `model(images, is_training)` form."""

    def __init__(self, cfg: ModelConfig, dtype=torch.float32):
        if dtype not in (torch.bfloat16, torch.float32):
            raise NotImplementedError("dtype must be torch.bfloat16 (the MFMA training path) or torch.float32 (exact fp32 arithmetic: every "
                                      "family's forward, loss and train step)")
        assert cfg.embed_dim % cfg.num_heads == 0  # vit.py:75
        self.cfg = cfg
        self.dtype = dtype
        self._engines: Dict[int, ViTEngine] = {}
        self._bound_id: Optional[Tuple[int, int]] = None  # (id(tree), version) last loaded into the engines

    # -- engine management: one engine per batch size, sharing parameter / gradient / bf16-weight buffers
    def _new_engine(self, batch: int):
        """The family's engine class (overridden by the subclasses below)."""
        if self.dtype == torch.float32:
            from .engine_f32 import ViTEngineF32

            return ViTEngineF32(self.cfg, batch)
        return ViTEngine(self.cfg, batch)

    def engine(self, batch: int) -> ViTEngine:
        e = self._engines.get(batch)
        if e is None:
            e = self._new_engine(batch)
            if self._engines:
                first = next(iter(self._engines.values()))
                e.params, e.grads, e.w = first.params, first.grads, first.w
                e.adam_m, e.adam_v = first.adam_m, first.adam_v
                e.weights_stale = first.weights_stale
            self._engines[batch] = e
        return e

    def _any_engine(self) -> ViTEngine:
        if not self._engines:
            return self.engine(1)
        return next(iter(self._engines.values()))

    # -- Flax-style protocol
    def init(self, rng, example_inputs: Optional[torch.Tensor] = None, is_training: bool = False) -> dict:
        """model.init(rng, ones(1,S,S,3), is_training=False) (train.py:29-31).  `rng` is an int seed.  The example
        input is used only to check the image size, as Flax uses it only for shapes."""
        if example_inputs is not None:
            S = self.cfg.img_size
            if tuple(example_inputs.shape[1:]) != (S, S, 3):
                raise ValueError(f"example input must be [B,{S},{S},3] (NHWC)")
        e = self._any_engine()
        e.init_params(int(rng))
        for o in self._engines.values():
            o.weights_stale = True
        return _clone_tree(e.param_tree())

    def bind(self, params: dict):
        """Load a Flax-shaped tree (torch tensors or numpy arrays) into the engine's flat HBM buffer."""
        e = self._any_engine()
        e.load_params(params)
        for o in self._engines.values():
            o.weights_stale = True
        return self

    def apply(self, params: dict, images: torch.Tensor, is_training: bool = False, rngs=None) -> torch.Tensor:
        """TrainState.apply_fn(params, images, is_training=...) (train.py:82,115).  images: NHWC [B,S,S,3] on the GPU,
        bf16 or fp32 (cast to bf16 like train.py:81).  ViT has no stochastic layer (all dropout rates are 0 in
        create_model.py:10-37), so is_training only selects nothing here and rngs is ignored."""
        self.bind(params)
        return self(images, is_training)

    def init_with_output(self, rng, inputs: torch.Tensor, is_training: bool = True):
        params = self.init(rng, inputs, is_training)
        return self(inputs, is_training), params

    def __call__(self, images: torch.Tensor, is_training: bool = False) -> torch.Tensor:
        if images.dim() != 4:
            raise ValueError("images must be NHWC [B,S,S,3]")
        e = self.engine(images.shape[0])
        logits = e.forward(images)
        return logits.to(self.dtype, copy=True)  # the engine owns (and reuses) its logits buffer

    # -- training conveniences used by train.py-shaped loops
    def parameters_tree(self) -> dict:
        return self._any_engine().param_tree()

    def gradients_tree(self) -> dict:
        return self._any_engine().grad_tree()


class CaiT(ViT):
    """models/cait.py:125-183 behind cait_engine.CaiTEngine.  `is_training=True` draws per-sample stochastic-depth masks
    (stochastic_depth.py:16-27) from the engine's generator; pass `rngs={'stochastic_depth': seed}` to seed it (the reference needs
    that rng stream too and forgets to pass it, defect B7)."""

    # NOTE: the reference computes CaiT in fp32 whatever dtype is passed (create_model.py:50-213 drop it)

    def _new_engine(self, batch: int):
        if self.dtype == torch.float32:
            from .engine_f32 import CaiTEngineF32

            return CaiTEngineF32(self.cfg, batch)
        from .cait_engine import CaiTEngine

        return CaiTEngine(self.cfg, batch)

    def apply(self, params: dict, images: torch.Tensor, is_training: bool = False, rngs=None) -> torch.Tensor:
        self.bind(params)
        e = self.engine(images.shape[0])
        if rngs and "stochastic_depth" in rngs:
            e.gen.manual_seed(int(rngs["stochastic_depth"]))
        return e.forward(images, is_training=is_training).to(self.dtype, copy=True)

    def __call__(self, images: torch.Tensor, is_training: bool = False) -> torch.Tensor:
        if images.dim() != 4:
            raise ValueError("images must be NHWC [B,S,S,3]")
        return self.engine(images.shape[0]).forward(images, is_training=is_training).to(self.dtype, copy=True)


class MLPMixer(ViT):
    """models/mlp_mixer.py:34-64 behind mixer_engine.MixerEngine.  No stochastic layer: is_training selects nothing."""

    def _new_engine(self, batch: int):
        if self.dtype == torch.float32:
            from .engine_f32 import MixerEngineF32

            return MixerEngineF32(self.cfg, batch)
        from .mixer_engine import MixerEngine

        return MixerEngine(self.cfg, batch)


class TNT(ViT):
    """models/tnt.py:136-193 behind tnt_engine.TNTEngine.  Every dropout rate is 0: is_training selects nothing."""

    def _new_engine(self, batch: int):
        if self.dtype == torch.float32:
            from .engine_f32 import TNTEngineF32

            return TNTEngineF32(self.cfg, batch)
        from .tnt_engine import TNTEngine

        return TNTEngine(self.cfg, batch)


def create_model(model_name: str, num_classes: int = 1000, dtype=torch.float32, img_size: int = 224):
    """models/create_model.py:6-8: same names, same default dtype (float32; train.py passes bfloat16).  `vit_ti_patch16` /
    `vit_s_patch16` added for BASELINE configs 1-2.  img_size is an extension (the reference fixes it through the init example;
    train.py --img_size)."""
    cfg = get_config(model_name, num_classes=num_classes, img_size=img_size)
    if cfg.kind == "vit":
        return ViT(cfg, dtype=dtype)
    if cfg.kind == "mixer":
        return MLPMixer(cfg, dtype=dtype)
    if cfg.kind == "tnt":
        return TNT(cfg, dtype=dtype)
    return CaiT(cfg, dtype=dtype)
