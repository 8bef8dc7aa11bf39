"""Every switch that changes an engine's launch plan, in one place.

Rounds 1-5 read 21 `SAVIT_*` environment variables at ~30 sites inside the engines (process-global, monkey-patched by tests).  They are
now fields of `EngineOptions`; an engine takes `options=` and / or keyword overrides, and the environment is read ONCE, at
construction, as the default of a field nobody set (so `SAVIT_CLS_ONLY_LAST=0 python bench.py` still works, and two engines in one
process can run different plans).  Precedence: keyword argument > `options=` object > environment > the default below.
`as_dict()` is what `bench.py` prints in its JSON line (`config.engine_options`), so a measured number names the plan it ran.

Nothing here is read by `libsavit.so` (the library reads no environment; tests/test_abi.py).
"""
from __future__ import annotations

import dataclasses
import os
from typing import Any, Dict, Mapping, Optional


def _flag(v: str) -> bool:
    return v != "0"


def _opt_flag(v: str) -> Optional[bool]:
    return None if v in ("", "auto") else v != "0"


def _opt_int(v: str) -> Optional[int]:
    return None if v in ("", "auto") else int(v)


@dataclasses.dataclass
class EngineOptions:
    # ---- data-parallel planning (ddp.py)
    reserved_cus: Optional[int] = None        # CUs left to a resident RCCL all-reduce during backward (None: 0)
    wgrad_max_lag: Optional[int] = None       # layers a weight gradient may wait for a full grouped launch (None: unbounded)
    # ---- the last ViT layer on the cls rows (round 5; exact)
    cls_only_last: bool = True                # backward of the last layer's MLP / LayerNorm / projection on B rows
    cls_fwd: bool = True                      # ... and its forward behind the qkv projection (cls-query attention kernels)
    rows_tile: bool = True                    # M <= 256 products on the few-rows kernel (tile 24); False: the LDS tiles
    # ---- step tail (round 5)
    first_touch: bool = True                  # grouped weight gradients store instead of accumulate (no gradient memset)
    defer_ln_finalize: bool = True            # one column-sum finalize launch per backward instead of one per LayerNorm
    wpe_grouped: bool = True                  # patch-embedding weight gradient as tiles of the last grouped launch
    wgrad_small_groups: bool = False          # a cls-only weight that found no slot as ONE grouped launch of its own
    # ---- weight-gradient scheduling
    wgrad_group: bool = True                  # tile FIFO of grouped launches (False: one launch per weight)
    wgrad_tile: Optional[int] = None          # tile code of the grouped launches (None: engine.wgrad_group_tile)
    wgrad_atomics: bool = False               # per-weight launches: fp32 atomics instead of slabs + ordered reduce
    overlap_wgrad: Optional[bool] = None      # weight gradients on a side stream (None: the engine family's default)
    side_streams: int = 1
    wgrad_cu_share: float = 0.56              # share of the CUs a side-stream weight-gradient launch is sized for
    ring_depth: int = 2                       # minimum depth of the cotangent rings
    # ---- family-specific
    th_fused: Optional[bool] = None           # CaiT: fused talking-heads kernels (None: what the library prefers for the geometry)
    tnt_seq16: bool = True                    # TNT: one-wave-per-sequence inner attention
    tnt_inner_splits: int = 160               # TNT: K-splits of the pixel-stream weight gradients

    ENV = {  # field -> (environment variable, parser)
        "reserved_cus": ("SAVIT_RESERVED_CUS", _opt_int), "wgrad_max_lag": ("SAVIT_WGRAD_MAX_LAG", _opt_int),
        "cls_only_last": ("SAVIT_CLS_ONLY_LAST", _flag), "cls_fwd": ("SAVIT_CLS_FWD", _flag), "rows_tile": ("SAVIT_ROWS_TILE", _flag),
        "first_touch": ("SAVIT_WGRAD_FIRST_TOUCH", _flag), "defer_ln_finalize": ("SAVIT_DEFER_LN_FINALIZE", _flag),
        "wpe_grouped": ("SAVIT_WPE_GROUPED", _flag), "wgrad_small_groups": ("SAVIT_WGRAD_SMALL_GROUPS", _flag),
        "wgrad_group": ("SAVIT_WGRAD_GROUP", lambda v: v not in ("0",)), "wgrad_tile": ("SAVIT_WGRAD_TILE", _opt_int),
        "wgrad_atomics": ("SAVIT_WGRAD_ATOMICS", lambda v: v == "1"), "overlap_wgrad": ("SAVIT_OVERLAP_WGRAD", _opt_flag),
        "side_streams": ("SAVIT_SIDE_STREAMS", int), "wgrad_cu_share": ("SAVIT_WGRAD_CU_SHARE", float),
        "ring_depth": ("SAVIT_RING_DEPTH", int), "th_fused": ("SAVIT_TH_FUSED", _opt_flag), "tnt_seq16": ("SAVIT_TNT_SEQ16", _flag),
        "tnt_inner_splits": ("SAVIT_TNT_INNER_SPLITS", int),
    }

    @classmethod
    def from_env(cls, env: Optional[Mapping[str, str]] = None, **overrides: Any) -> "EngineOptions":
        """Defaults, then the environment, then `overrides` (None = not given)."""
        env = os.environ if env is None else env
        kw: Dict[str, Any] = {}
        for field, (name, parse) in cls.ENV.items():
            v = env.get(name)
            if v is not None and v != "":
                kw[field] = parse(v)
        o = cls(**kw)
        return o.replace(**overrides)

    def replace(self, **overrides: Any) -> "EngineOptions":
        unknown = set(overrides) - {f.name for f in dataclasses.fields(self)}
        if unknown:
            raise TypeError(f"unknown engine option(s): {sorted(unknown)}")
        return dataclasses.replace(self, **{k: v for k, v in overrides.items() if v is not None})

    @classmethod
    def resolve(cls, options: Optional["EngineOptions"] = None, **overrides: Any) -> "EngineOptions":
        """What an engine constructor calls: `options` (or the environment's view when None) with the explicit keywords on top."""
        base = options if options is not None else cls.from_env()
        return base.replace(**overrides)

    def as_dict(self) -> Dict[str, Any]:
        return dataclasses.asdict(self)

    def non_default(self) -> Dict[str, Any]:
        d0 = EngineOptions()
        return {k: v for k, v in self.as_dict().items() if getattr(d0, k) != v}
