"""ViT training engine: owns the HBM layout (flat fp32 parameters / gradients, bf16 MFMA operand copies,
per-layer saved activations) and the fixed launch sequence of forward, loss and backward.

Reference path being replaced: the pmapped `train_step` of /root/reference/train.py:77-100 around
`ViT.__call__` (models/vit.py:73-99): forward, label-smoothed CE, reverse-mode AD.  Everything that touches
tensors is a C-ABI kernel call (lib.py); PyTorch only allocates device memory and provides the stream.

HBM layout
  params / grads : ONE fp32 buffer each, layer-major
      [embed: Wpe | cls | pos] [layer 0: ln1 g,b | Wqkv | Wo | ln2 g,b | W1 | b1 | W2 | b2] ... [final: lnf g,b | Wh | bh]
    so the optimizer is a single launch and gradient buckets for the data-parallel all-reduce are contiguous
    slices that become final in reverse order during backward.  Kernels are Flax-layout [in, out]; q/k/v
    kernels are fused into Wqkv [d, 3d] (the Flax tree exposes strided views).
  wbf : bf16 copies of every matrix in BOTH layouts ([in,out] for input-gradient GEMMs, [out,in] for forward
    GEMMs - all GEMMs are "TN"), refreshed by 6 batched cast/transposes after each optimizer step.
  activations : per layer x_in, x_mid (fp32 residual stream), h1, h2, o, qkv, u, a (bf16), LN stats, LSE.

The launch plan (ctypes function + prebuilt argument tuple per kernel) is built once per batch size, so a step is
a flat loop of ~260 asynchronous launches with no allocation - and therefore capturable in a hipGraph.
"""
from __future__ import annotations

import ctypes
import os
import math
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import lib as _lib
from .config import ModelConfig
from .options import EngineOptions
from .timing import timed_call

bf16 = torch.bfloat16
f32 = torch.float32


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


class ParamLayout:
    """Offsets (in fp32 elements) of every tensor in the flat parameter buffer."""

    def __init__(self, cfg: ModelConfig):
        if cfg.kind != "vit":
            raise NotImplementedError("ParamLayout: only the ViT family is laid out here")
        self.cfg = cfg
        d, F, C, N, L = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers
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
        add("cls", (d,))
        add("pos", (N, d))
        cur = _align(cur, 64)
        self.embed_end = cur
        self.layer_start: List[int] = []
        for l in range(L):
            self.layer_start.append(cur)
            add(f"l{l}.ln1_g", (d,))
            add(f"l{l}.ln1_b", (d,))
            add(f"l{l}.Wqkv", (d, 3 * d))
            add(f"l{l}.Wo", (d, d))
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
        o, shape = self.off[name]
        n = 1
        for s in shape:
            n *= s
        return flat[o:o + n].view(*shape)

    def flax_tree(self, flat: torch.Tensor) -> dict:
        """Flax-shaped nested dict of VIEWS into `flat`."""
        cfg = self.cfg
        d, H, hd = cfg.embed_dim, cfg.num_heads, cfg.head_dim
        v = lambda n: self.view(flat, n)  # noqa: E731
        enc = {"AddAbsPosEmbed_0": {"pos_embed": v("pos").view(1, cfg.seq_len, d)}}
        for l in range(cfg.num_layers):
            wqkv = v(f"l{l}.Wqkv")
            enc[f"EncoderBlock_{l}"] = {
                "LayerNorm_0": {"scale": v(f"l{l}.ln1_g"), "bias": v(f"l{l}.ln1_b")},
                "SelfAttentionBlock_0": {
                    "queries": {"kernel": wqkv[:, 0:d].unflatten(1, (H, hd))},
                    "keys": {"kernel": wqkv[:, d:2 * d].unflatten(1, (H, hd))},
                    "values": {"kernel": wqkv[:, 2 * d:3 * d].unflatten(1, (H, hd))},
                    "DenseGeneral_0": {"kernel": v(f"l{l}.Wo").view(H, hd, d)},
                },
                "LayerNorm_1": {"scale": v(f"l{l}.ln2_g"), "bias": v(f"l{l}.ln2_b")},
                "FFBlock_0": {"Dense_0": {"kernel": v(f"l{l}.W1"), "bias": v(f"l{l}.b1")},
                              "Dense_1": {"kernel": v(f"l{l}.W2"), "bias": v(f"l{l}.b2")}},
            }
        enc["LayerNorm_0"] = {"scale": v("lnf_g"), "bias": v("lnf_b")}
        return {"params": {
            "PatchEmbedBlock_0": {"Dense_0": {"kernel": v("Wpe")}},
            "cls": v("cls").view(1, 1, d),
            "Encoder_0": enc,
            "Dense_0": {"kernel": v("Wh"), "bias": v("bh")},
        }}


def _copy_tree(dst: dict, src: dict, path: str = ""):
    for k, dv in dst.items():
        if k not in src:
            raise KeyError(f"missing parameter {path}/{k}")
        sv = src[k]
        if isinstance(dv, dict):
            _copy_tree(dv, sv, f"{path}/{k}")
        else:
            if not isinstance(sv, torch.Tensor) and hasattr(sv, "flags") and not sv.flags.writeable:
                sv = sv.copy()  # e.g. arrays restored from a checkpoint buffer (np.frombuffer views are read-only)
            t = torch.as_tensor(sv) if not isinstance(sv, torch.Tensor) else sv
            if tuple(t.shape) != tuple(dv.shape):
                raise ValueError(f"{path}/{k}: shape {tuple(t.shape)} != {tuple(dv.shape)}")
            dv.copy_(t.to(device=dv.device, dtype=dv.dtype))
    extra = set(src) - set(dst)
    if extra:
        raise KeyError(f"unexpected parameters under {path}: {sorted(extra)}")


class _Plan:
    """A recorded sequence of kernel launches: (ctypes fn, args-without-stream, label).

    A launch may be marked `side`: it has no consumer later in the plan (the weight-gradient GEMMs: their only output is
    the gradient buffer), so `run_overlapped` issues it on one of several extra HIP streams, round-robin, where it runs
    BESIDE the main chain: those launches are sized for about half the CUs (few K-splits, so also few fp32 atomics), the
    main chain keeps the rest.  `reads`/`writes` declare the scratch buffers a side launch reads and a main launch
    overwrites; the plan turns them into event waits."""

    def __init__(self):
        self.calls: List[Tuple[Callable, tuple, str]] = []
        self.keep: List[object] = []  # ctypes structs that must outlive the plan
        self.ws_requests: List[Tuple[int, int, int, int]] = []  # (call index, workspace key, bytes, argument position): add_wgrad
        self.hook_alias: Dict[str, List[str]] = {}  # launch label -> labels whose hooks fire behind THIS launch instead (add_wgrad, wgrad groups)
        self.side: Dict[int, int] = {}           # call index -> side-launch ordinal (stream = ordinal % number of side streams)
        self.guard: Dict[int, List[int]] = {}    # main call index -> side call indices that must have finished first
        self._readers: Dict[int, List[int]] = {}  # buffer address -> side calls reading it (build-time bookkeeping)
        self._ev_ready: Dict[int, "torch.cuda.Event"] = {}
        self._ev_done: Dict[int, "torch.cuda.Event"] = {}
        # first-touch weight gradients (round 5): element ranges of the gradient buffer that the plan's grouped launches store completely
        # (savit_wgrad_problem.overwrite), the problem arrays holding those flags, what is left to zero / to sum squares over, and whether
        # the launches also accumulate the sum of squares of what they store (engine.gnorm[1:33])
        self.covered: Dict[int, int] = {}
        self.wgrad_arrays: List[object] = []
        self.rest_ranges = None   # ctypes c_long array of (offset, length) pairs, or None: the whole buffer
        self.n_rest = 0
        self.fold_sumsq = False

    def add(self, fn, args: tuple, label: str, side: bool = False, reads: tuple = (), writes: tuple = (), same_side_stream: bool = False):
        i = len(self.calls)
        self.calls.append((fn, args, label))
        if side:
            assert not writes
            # same_side_stream: this launch continues the previous side launch (same ordinal -> same stream, issued right behind it)
            self.side[i] = self.side[max(self.side)] if (same_side_stream and self.side) else (max(self.side.values()) + 1 if self.side else 0)
            for r in reads:
                self._readers.setdefault(r, []).append(i)
        else:
            g = sorted({j for w in writes for j in self._readers.pop(w, [])})
            if g:
                self.guard[i] = g

    def hook_for(self, hooks, label: str):
        """The hook to run right behind the launch `label`: a label that has an alias target fires there, not here."""
        if not hooks:
            return None
        names = self.hook_alias.get(label)
        if names is None:
            if any(label in v for v in self.hook_alias.values()):
                return None  # deferred: fires behind another launch
            return hooks.get(label)
        cbs = [hooks[n] for n in names if n in hooks]
        if not cbs:
            return None
        if len(cbs) == 1:
            return cbs[0]

        def fire():
            for cb in cbs:
                cb()
        return fire

    def run(self, stream: int, timer=None, hooks: Optional[Dict[str, Callable[[], None]]] = None):
        """Issue every launch on `stream`.  timer (timing.LaunchTimer): bracket the launches it tracks; hooks: run the callback of a
        label (DDP bucket trigger) right behind its launch."""
        if timer is None and not hooks:
            for fn, args, label in self.calls:
                rc = fn(*args, stream)
                if rc != 0:
                    _lib.check(rc, label)
            return
        for fn, args, label in self.calls:
            k = timer.begin(label, stream) if timer is not None else -1
            rc = fn(*args, stream)
            if k >= 0:
                timer.end(k, stream)
            if rc != 0:
                _lib.check(rc, label)
            cb = self.hook_for(hooks, label)
            if cb is not None:
                cb()

    def run_overlapped(self, main: "torch.cuda.Stream", sides: List["torch.cuda.Stream"],
                       hooks: Optional[Dict[str, Callable[[], None]]] = None, timer=None):
        """Main-chain launches on `main`, side launches round-robin on `sides`, ordered by events; every stream is joined
        before each hook (the DDP bucket all-reduce reads gradients written on any of them) and at the end.  timer: as in run()
        (a side launch is bracketed on ITS stream: beside the main chain its wall time includes the time it shares the CUs)."""
        mh = main.cuda_stream
        ns = len(sides)
        last = [None] * ns  # last side call issued per side stream
        for i, (fn, args, label) in enumerate(self.calls):
            k = self.side.get(i)
            if k is not None:
                st = sides[k % ns]
                ev = self._ev_ready.get(i)
                if ev is None:
                    ev = self._ev_ready[i] = torch.cuda.Event()
                    self._ev_done[i] = torch.cuda.Event()
                ev.record(main)
                st.wait_event(ev)
                tk = timer.begin(label, st.cuda_stream) if timer is not None else -1
                rc = fn(*args, st.cuda_stream)
                if tk >= 0:
                    timer.end(tk, st.cuda_stream)
                self._ev_done[i].record(st)
                last[k % ns] = i
            else:
                for j in self.guard.get(i, ()):
                    main.wait_event(self._ev_done[j])
                tk = timer.begin(label, mh) if timer is not None else -1
                rc = fn(*args, mh)
                if tk >= 0:
                    timer.end(tk, mh)
            if rc != 0:
                _lib.check(rc, label)
            cb = self.hook_for(hooks, label)
            if cb is not None:
                for j in last:
                    if j is not None:
                        main.wait_event(self._ev_done[j])
                cb()
        for j in last:
            if j is not None:
                main.wait_event(self._ev_done[j])


def add_wgrad(eng, plan: _Plan, label: str, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, splits: int, patch=(0, 0, 0, 0), side: bool = True):
    """Record a weight-gradient launch of any engine of this package (savit_gemm_bf16_wgrad_ws: the splits of the reduction over the
    tokens go through partial slabs + an ordered sum - no atomics, bitwise reproducible).  side launches have no consumer inside
    backward, `_Plan.run_overlapped` puts them on a side stream.  The slab workspace is attached by `finalize_wgrad_ws` once the
    whole plan is known: launches that can run at the same time must not share one, so side launches use the workspace of THEIR
    side stream (ordinal % number of side streams, the stream run_overlapped will pick) and main-chain launches a separate one."""
    need = int(eng.L.savit_gemm_wgrad_workspace_bytes(Mr, Kin, Nout, splits, patch[0]))
    if eng.opt.wgrad_atomics:
        need = 0
    on_side = side and eng.overlap_wgrad and not eng._building_serial
    key = ((max(plan.side.values()) + 1 if plan.side else 0) % max(1, eng.n_side_streams)) if on_side else -1
    if need <= 0:  # small / ragged shapes served by the 2-stage kernel: fp32 atomics, one launch
        plan.add(eng.L.savit_gemm_bf16_wgrad, (X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, splits, patch[0], patch[1], patch[2], patch[3]),
                 label, side=side, reads=(dY,) if side else ())
        return
    nsplit = int(eng.L.savit_gemm_wgrad_split_count(Mr, Kin, Nout, splits, patch[0]))
    # two plan entries = two kernels (the GEMM that stores the split partials, the ordered sum into dW), the second on the SAME
    # stream as the first
    plan.add(eng.L.savit_gemm_bf16_wgrad_partial, (X, dY, Mr, Kin, Nout, ldx, lddy, splits, patch[0], patch[1], patch[2], patch[3], None, 0),
             label, side=side, reads=(dY,) if side else ())
    plan.ws_requests.append((len(plan.calls) - 1, key, need, -2))
    plan.add(eng.L.savit_gemm_wgrad_reduce, (None, nsplit, Kin, Nout, dW, lddw), label + ".reduce", side=side, same_side_stream=True)
    plan.ws_requests.append((len(plan.calls) - 1, key, need, 0))
    plan.hook_alias[label + ".reduce"] = [label]  # a hook registered for `label` (DDP bucket trigger) fires behind the reduce


def add_wgrad_group(eng, plan: _Plan, label: str, entries: list, tile: int, deferred_hooks: list, side: bool = True):
    """Record ONE grouped weight-gradient launch (savit_gemm_bf16_wgrad_grouped).  entries = [(problem, tile_begin, tile_count)] with
    problem = (X, dY, dW, M, Kin, Nout, ldx, lddy, lddw): one workgroup per output tile over all tokens - no split, no slabs, no reduce
    launch.  `deferred_hooks`: labels of earlier launches whose hooks (DDP bucket triggers) must wait for these gradients; they fire
    behind this launch."""
    arr = (_lib.WgradProblem * len(entries))()
    flops = 0.0
    g0, g1 = eng.grads.data_ptr(), eng.grads.data_ptr() + eng.grads.numel() * 4
    first_touch = bool(getattr(eng, "first_touch", False))
    for q, ((X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw), t0, tc) in zip(arr, entries):
        q.X, q.dY, q.dW, q.M, q.Kin, q.Nout, q.ldx, q.lddy, q.lddw, q.tile_begin, q.tile_count = X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, t0, tc
        flops += 2.0 * Mr * Kin * Nout * tc / int(eng.L.savit_gemm_wgrad_group_tiles(Kin, Nout, tile))  # the weight's flops, by its share of tiles in this launch
        # every tile of a queued weight goes out in SOME grouped launch of this plan, each written by exactly one workgroup: a dense
        # matrix inside the gradient buffer is therefore stored, not accumulated, and needs no memset (first touch)
        if first_touch and lddw == Nout and g0 <= dW and dW + Kin * Nout * 4 <= g1 and (dW - g0) % 16 == 0 and (Kin * Nout) % 4 == 0:
            q.overwrite = 1
            plan.covered[(dW - g0) // 4] = Kin * Nout
    plan.keep.append(arr)
    plan.wgrad_arrays.append(arr)
    fold = first_touch and getattr(eng, "_fold_sumsq_ok", False)
    if fold and not all(q.overwrite for q in arr):
        # the kernel publishes the squares of EVERY problem of a launch; an entry that failed the coverage predicate is also in
        # `rest_ranges` (finalize_first_touch) and would be counted twice in the clip norm.  No ViT / CaiT weight fails it
        # today; a launch that ever mixes the two kinds accumulates all of its entries (they are then zeroed and summed with the rest)
        for q in arr:
            if q.overwrite:
                q.overwrite = 0
                plan.covered.pop((q.dW - g0) // 4, None)
        fold = False
    slots = eng.gnorm.data_ptr() + 4 if fold else None
    plan.fold_sumsq = bool(plan.fold_sumsq or slots is not None)  # any launch publishing squares: the norm = fold slots + rest_ranges
    plan.add(eng.L.savit_gemm_bf16_wgrad_grouped_ex, (arr, len(entries), tile, slots), label, side=side,
             reads=tuple(sorted({e[0][1] for e in entries})) if side else ())
    if deferred_hooks:
        plan.hook_alias[label] = list(deferred_hooks)
    if not hasattr(eng, "group_flops"):
        eng.group_flops = {}
    eng.group_flops[label] = flops


def wgrad_group_tile(d: int, F: int, override: Optional[int] = None) -> int:
    """Tile code of the grouped weight-gradient launches for a model of width d / hidden width F: 256 x 256 where the matrices are
    multiples of it (DeiT-B, ViT-L); for the d = 384 models (DeiT-S, CaiT-S) 256 x 384 or 384 x 256 per weight, whichever covers it with
    fewer tiles (95 % of a launch inside a matrix; with 256 x 256 tiles 29 % hung over the matrix edges); else 256 x 256 with edge tiles."""
    if override:  # A/B runs (EngineOptions.wgrad_tile)
        return int(override)
    if d % 256 == 0 and F % 256 == 0:
        return 256
    if d % 384 == 0 and F % 384 == 0:
        return 640  # 256 x 384 or 384 x 256 per weight (19 tiles per layer at d = 384; 128 x 384 tiles: code 384, 36 per layer)
    return 256


class WgradQueue:
    """FIFO of weight-gradient output tiles waiting for a grouped launch.  A launch takes exactly `cap` tiles (one per CU the rank
    may count on: a full round of workgroups), cutting a weight between two launches where it must; what is left at the end goes out
    as the last launch.  max_lag (layers, None = unbounded): a launch also goes out, as a partial round, once the oldest waiting
    gradient is that many layers behind - it bounds how far a data-parallel bucket trigger is deferred and how deep the cotangent rings
    must be (narrow models fill a round only every 6-15 layers)."""

    MAX_ENTRIES = 64  # savit_gemm_bf16_wgrad_grouped takes at most this many (weight, tile range) entries per launch

    def __init__(self, cap: int, max_lag: Optional[int] = None):
        self.cap = cap
        self.max_lag = max_lag
        self.items: List[list] = []  # [problem, layer, tiles, next tile]

    def push(self, problem, layer: int, tiles: int):
        self.items.append([problem, layer, tiles, 0])

    def pending(self) -> int:
        return sum(it[2] - it[3] for it in self.items)

    def due(self, layer: Optional[int] = None) -> bool:
        """A launch's worth is waiting: a full round of tiles, as many entries as one launch takes (narrow models), or - with
        max_lag - a gradient that has waited max_lag layers (backward is at `layer`, counting down)."""
        if self.pending() >= self.cap or len(self.items) >= self.MAX_ENTRIES:
            return True
        return self.max_lag is not None and layer is not None and bool(self.items) and self.items[0][1] - layer >= self.max_lag

    def take(self, n: int):
        """-> (entries [(problem, tile_begin, tile_count)], layers whose LAST pending tile is in this launch, oldest layer touched)"""
        entries, done, oldest = [], [], None
        while n > 0 and self.items and len(entries) < self.MAX_ENTRIES:
            it = self.items[0]
            c = min(n, it[2] - it[3])
            entries.append((it[0], it[3], c))
            oldest = it[1] if oldest is None else max(oldest, it[1])
            it[3] += c
            n -= c
            if it[3] == it[2]:
                self.items.pop(0)
                if not any(o[1] == it[1] for o in self.items):
                    done.append(it[1])
        return entries, done, oldest


def finalize_first_touch(eng, plan: _Plan):
    """After a backward plan is recorded: the ranges of the gradient buffer its grouped launches do NOT store (everything else must be
    zeroed before backward and summed for the gradient norm after it).  Without covered ranges the plan keeps the whole-buffer forms."""
    total = eng.grads.numel()
    if not plan.covered:
        plan.rest_ranges, plan.n_rest, plan.fold_sumsq = None, 0, False
        return
    rest, cur = [], 0
    for off in sorted(plan.covered):
        if off > cur:
            rest.append((cur, off - cur))
        cur = max(cur, off + plan.covered[off])
    if cur < total:
        rest.append((cur, total - cur))
    assert all(o % 4 == 0 and n % 4 == 0 for o, n in rest), "gradient ranges are 16-byte aligned (ParamLayout aligns every tensor)"
    arr = (ctypes.c_long * (2 * len(rest)))(*[v for r in rest for v in r])
    plan.keep.append(arr)
    plan.rest_ranges, plan.n_rest = arr, len(rest)
    plan.rest_elems = sum(n for _, n in rest)


def finalize_wgrad_ws(eng, plan: _Plan):
    """Allocate (or grow) the engine's slab workspaces to the largest request of `plan` per key and patch its launches."""
    if not hasattr(eng, "_wgrad_wsbuf"):
        eng._wgrad_wsbuf, eng._wgrad_ws_old = {}, []
    for key in sorted({k for _, k, _, _ in plan.ws_requests}):
        need = max(n for _, k, n, _ in plan.ws_requests if k == key)
        buf = eng._wgrad_wsbuf.get(key)
        if buf is None or buf.numel() < need:
            if buf is not None:
                eng._wgrad_ws_old.append(buf)  # an earlier plan still points at it
            eng._wgrad_wsbuf[key] = torch.empty(need, dtype=torch.uint8, device=eng.dev)
    for idx, key, _, pos in plan.ws_requests:  # pos: index of the workspace pointer in the argument tuple (-2: pointer, then its size)
        fn, args, label = plan.calls[idx]
        buf = eng._wgrad_wsbuf[key]
        args = list(args)
        args[pos] = buf.data_ptr()
        if pos == -2:
            args[-1] = buf.numel()
        plan.calls[idx] = (fn, tuple(args), label)
    plan.ws_requests = []


class ViTEngine:
    DEFAULT_OVERLAP = False  # weight gradients on a side stream? (see _init_step_state)

    def __init__(self, cfg: ModelConfig, batch: int, device: str = "cuda", round_like_reference: bool = True,
                 reserved_cus: Optional[int] = None, wgrad_max_lag: Optional[int] = None, options: Optional[EngineOptions] = None, **opts):
        """reserved_cus: CUs this rank leaves to a resident RCCL all-reduce (train.py:96) - grids are then planned for the remaining
        ones (default: SAVIT_RESERVED_CUS, else 0; ddp.default_reserved_cus(world) is what bench.py / train.py pass at world > 1).
        wgrad_max_lag: bound, in layers, on how long a weight gradient waits for a full grouped launch (WgradQueue).
        options / **opts: every other plan switch (options.EngineOptions: cls_only_last, cls_fwd, rows_tile, first_touch, overlap_wgrad ...);
        a field nobody sets takes its SAVIT_* environment variable, read here, once."""
        self.opt = EngineOptions.resolve(options, reserved_cus=reserved_cus, wgrad_max_lag=wgrad_max_lag, **opts)
        if cfg.kind != "vit":
            raise NotImplementedError("ViTEngine handles the ViT family; CaiT uses CaiTEngine")
        if cfg.head_dim not in (48, 64):
            raise NotImplementedError("attention kernels are built for head_dim 48 and 64")
        # (any sequence length: up to 608 tokens one head's K / V stay resident in LDS, longer sequences stream them - csrc/attention.hip)
        if cfg.embed_dim % 64 != 0 or cfg.patch % 8 != 0 or cfg.num_classes % 8 != 0:
            raise ValueError("embed_dim % 64, patch % 8 and num_classes % 8 must be 0")
        if not torch.cuda.is_available():
            raise RuntimeError("ViTEngine needs a GPU: there is no CPU path")
        self.L = _lib.load()
        self.cfg = cfg
        self.B = int(batch)
        self.dev = torch.device(device)
        self.rp = int(round_like_reference)
        self._init_cu_budget()
        self.layout = ParamLayout(cfg)
        d, F, C, N, NL = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers
        self.M = self.B * N
        self.Cp = _align(C, 64)
        z = lambda *s, dt=f32: torch.zeros(*s, dtype=dt, device=self.dev)  # noqa: E731
        e = lambda *s, dt=f32: torch.empty(*s, dtype=dt, device=self.dev)  # noqa: E731
        self._init_flat_buffers()
        # ---- bf16 operand copies.  The [in, out] forms (input-gradient GEMMs) are views into a flat bf16 MIRROR of the parameter
        # buffer that the fused AdamW writes together with the fp32 parameters (savit_adamw_step_mirror); the [out, in] forms (forward
        # GEMMs) are transposed out of that mirror, 2 B read per element instead of 4 (round 4: 0.22 -> 0.1 ms per step).
        self.params_bf16 = torch.zeros(self.layout.total, dtype=bf16, device=self.dev)
        self._mirror_fresh = False  # True right after an optimizer step (the mirror already holds bf16(params))
        lay = self.layout
        for l in range(NL):
            for nm in ("Wqkv", "Wo", "W1", "W2"):
                assert lay.off[f"l{l}.{nm}"][0] % 8 == 0, "16-byte aligned bf16 operand views"
        mview = lambda nm, R, Cc: [self.params_bf16[lay.off[f"l{l}.{nm}"][0]:lay.off[f"l{l}.{nm}"][0] + R * Cc].view(R, Cc) for l in range(NL)]  # noqa: E731
        self.w = {
            "Wqkv_n": mview("Wqkv", d, 3 * d), "Wqkv_t": e(NL, 3 * d, d, dt=bf16),
            "Wo_n": mview("Wo", d, d), "Wo_t": e(NL, d, d, dt=bf16),
            "W1_n": mview("W1", d, F), "W1_t": e(NL, F, d, dt=bf16),
            "W2_n": mview("W2", F, d), "W2_t": e(NL, d, F, dt=bf16),
            "Wpe_t": e(d, cfg.patch_dim, dt=bf16),
            "Wh_t": e(C, d, dt=bf16), "Wh_n": z(d, self.Cp, dt=bf16),
        }
        # ---- activations (saved for backward)
        M = self.M
        self.x = [e(M, d) for _ in range(NL + 1)]      # residual stream entering layer l (x[NL] = encoder output)
        self.xmid = [e(M, d) for _ in range(NL)]
        self.h1 = [e(M, d, dt=bf16) for _ in range(NL)]
        self.h2 = [e(M, d, dt=bf16) for _ in range(NL)]
        self.qkv = [e(M, 3 * d, dt=bf16) for _ in range(NL)]
        self.o = [e(M, d, dt=bf16) for _ in range(NL)]
        self.u = [e(M, F, dt=bf16) for _ in range(NL)]
        self.a = [e(M, F, dt=bf16) for _ in range(NL)]
        self.stats = [e(4, M) for _ in range(NL)]      # mean1, rstd1, mean2, rstd2
        self.lse = [e(self.B, cfg.num_heads, N) for _ in range(NL)]
        self.zcls = e(self.B, d, dt=bf16)
        self.fstats = e(2, self.B)
        # ---- backward scratch
        self.dres = e(M, d)
        # Weight gradients wait in a FIFO of output tiles and go out in grouped launches of exactly one tile per CU (add_wgrad_group,
        # WgradQueue), up to `wgrad_lag` layers after their cotangents were produced: those cotangent buffers therefore rotate through
        # rings that deep (also what lets side-stream weight-gradient GEMMs lag behind the main chain).
        self.wgrad_tile, self.wgrad_cap, self.wgrad_divert, self.wgrad_lag = self._wgrad_group_plan()
        self.wgrad_group = self.wgrad_lag  # (layers a launch reaches back: bench.py's label for the grouping)
        depth = max(2, self.opt.ring_depth, self.wgrad_lag + 1 if self.wgrad_tile else 0)
        self.dres_b_ring = [e(M, d, dt=bf16) for _ in range(2 * depth)]
        self.dres_b = self.dres_b_ring[0]
        self.d_u_ring = [e(M, F, dt=bf16) for _ in range(depth)]
        self.d_u = self.d_u_ring[0]
        self.d_h = e(M, d, dt=bf16)
        self.d_o = e(M, d, dt=bf16)
        self.dqkv_ring = [e(M, 3 * d, dt=bf16) for _ in range(depth)]
        self.dqkv = self.dqkv_ring[0]
        self.colsum_slab = e(max(1, self.L.savit_gemm_colsum_rows_cus(M, F, d, 0, self.cu_budget if self.reserved_cus else 0)), F)
        self.d_z = e(self.B, d, dt=bf16)
        ws = self.L.savit_layernorm_bwd_workspace_bytes(M, d)
        self.ln_ws = torch.empty(max(int(ws), 16), dtype=torch.uint8, device=self.dev)
        self._init_step_state()

    # ---- state every engine of this package shares (the Mixer / TNT engines subclass this one and lay out their own activations)
    def _init_cu_budget(self):
        """CUs the launch plans may count on.  Everything that sizes a grid for "one round of workgroups" reads cu_budget: the grouped
        weight-gradient launches (WgradQueue cap), the TN GEMM tile choice (savit_gemm_args.cu_budget) and the persistent attention
        kernels' grids (savit_set_cu_budget)."""
        self.n_cus = torch.cuda.get_device_properties(self.dev).multi_processor_count
        reserved_cus = self.opt.reserved_cus or 0
        if not 0 <= int(reserved_cus) < self.n_cus:
            raise ValueError(f"reserved_cus must be in [0, {self.n_cus})")
        self.reserved_cus = int(reserved_cus)
        self.cu_budget = self.n_cus - self.reserved_cus
        # the persistent attention kernels size their grids (one workgroup per CU) from the same budget.  The library's setting is
        # process-wide, so it is raised around THIS engine's backward launches only (`_run_bwd`): forward and optimizer have no
        # all-reduce beside them, and another engine in the process is not affected.
        self._building_bwd = False  # set while a backward plan is recorded: only its launches run beside the all-reduce
        # Round 5: behind the final LayerNorm only the cls rows carry a gradient (vit.py:57,95: row 0 alone reaches the head), so the LAST
        # encoder layer's MLP branch, its second LayerNorm and its output projection are differentiated on the B cls rows instead of the
        # B*N token rows - every other row of those cotangents is exactly zero (ViTEngine._record_bwd_plan; wide models only)
        cfg_ = getattr(self, "cfg", None)
        self.cls_only_last = (self.opt.cls_only_last and getattr(cfg_, "kind", "") == "vit" and
                              cfg_.embed_dim > 64 and cfg_.seq_len > 1)
        # ... and (cls_fwd) its FORWARD behind the qkv projection as well - the cls query's attention (savit_cls_query_attention_fwd), output
        # projection, second LayerNorm and MLP on B rows: the other rows of the last layer's output are never read (the final LayerNorm and
        # the head take row 0), and its backward then works on compact [B, .] activations.  Needs the cls-query kernels' geometry.
        self.cls_fwd = bool(self.cls_only_last and self.opt.cls_fwd and cfg_.head_dim in (48, 64) and cfg_.seq_len <= 640)
        self._needs_zero_dres = True
        self._cls_per_weight = frozenset({"W2", "W1", "Wo"})  # (_wgrad_group_plan moves the ones that fit the last round into the tile FIFO)
        self.wgrad_max_lag = self.opt.wgrad_max_lag

    def _init_flat_buffers(self):
        """Parameters, gradients, optimizer state: flat fp32 buffers in the layout's order (Adam moments allocated on first use)."""
        self.params = torch.zeros(self.layout.total, dtype=f32, device=self.dev)
        self.grads = torch.zeros(self.layout.total, dtype=f32, device=self.dev)
        self.adam_m = None
        self.adam_v = None
        self.step_count = 0
        # gnorm[0]: the squared global gradient norm the fused AdamW reads; gnorm[1:33]: the accumulators the grouped weight-gradient
        # launches add the squares of what they store to (savit_gemm_bf16_wgrad_grouped_ex)
        self.gnorm = torch.zeros(36, dtype=f32, device=self.dev)
        self.gnorm_sq = self.gnorm[0:1]

    def _init_step_state(self):
        """Loss I/O, the bf16 NHWC input buffer, launch plans, DDP hooks and the side-stream settings."""
        cfg, B = self.cfg, self.B
        z = lambda *s, dt=f32: torch.zeros(*s, dtype=dt, device=self.dev)  # noqa: E731
        self.logits = torch.empty(B, cfg.num_classes, dtype=f32, device=self.dev)
        self.dlogits = z(B, self.Cp, dt=bf16)
        self.labels = torch.zeros(B, dtype=torch.int32, device=self.dev)
        self.loss = z(1)
        self.loss_rows = z(B)
        self.top1 = z(B)
        self.top5 = z(B)
        self.images: Optional[torch.Tensor] = None  # bf16 NHWC, set by forward()
        self._img_buf = torch.empty(B, cfg.img_size, cfg.img_size, 3, dtype=bf16, device=self.dev)
        self._fwd_plan: Optional[_Plan] = None
        self._bwd_plan: Optional[_Plan] = None
        self._cast_plan: Optional[_Plan] = None
        self._bwd_hooks: Dict[str, Callable[[], None]] = {}  # label -> callback run right after that launch (DDP buckets): `bwd_hooks`
        self._data_parallel = False  # sticky: set once hooks are attached (the plans of a data-parallel rank differ, see bwd_hooks)
        # Round 5: grouped weight gradients store by first touch (no 346 MB memset, no read-modify-write); alone on
        # the GPU they also carry the gradient norm's sum of squares, and the LayerNorm backward launches leave their column-sum slabs
        # to ONE finalize launch at the end of backward.  A data-parallel rank keeps the per-launch finalizes and the separate norm
        # pass: its bucket triggers need final bias / LayerNorm gradients layer by layer, and its norm is that of the REDUCED gradient.
        self.first_touch = self.opt.first_touch
        self.defer_ln_finalize = self.opt.defer_ln_finalize
        self._gnorm_folded = False   # True between a backward whose launches accumulated gnorm[1:33] and the optimizer step that uses them
        self._accumulate_run = False
        self.launch_timer = None  # timing.LaunchTimer: brackets the launches it tracks (bench.py, profile_step)
        self.weights_stale = True
        # weight-gradient GEMMs on a second stream (SAVIT_OVERLAP_WGRAD=1 / 0 overrides the engine's default).  Round 2: with the
        # atomic-free weight gradients and the tail-split tiles the ViT family runs as fast or faster on ONE stream (DeiT-B 6 715 vs
        # 6 650 img/s, DeiT-S 18 050 vs 17 980, same box) - the second stream only ever filled bubbles those changes removed - while
        # the Mixer (+5 %) and TNT (+12 %) engines, with their many small launches, still gain from it (DEFAULT_OVERLAP below).
        self.overlap_wgrad = self.DEFAULT_OVERLAP if self.opt.overlap_wgrad is None else bool(self.opt.overlap_wgrad)
        self.n_side_streams = int(self.opt.side_streams)
        self.wgrad_cu_share = float(self.opt.wgrad_cu_share)
        self._side_streams: List[torch.cuda.Stream] = []
        self._building_serial = False
        self._bwd_plan_serial: Optional[_Plan] = None  # every launch sized for the whole chip: profile_step / one-stream runs

    @property
    def bwd_hooks(self) -> Dict[str, Callable[[], None]]:
        return self._bwd_hooks

    @bwd_hooks.setter
    def bwd_hooks(self, hooks):
        """Attaching hooks (ddp.GradSync.hooks()) puts the engine in data-parallel planning mode for good: backward plans recorded
        before are dropped (they may defer column sums past a bucket trigger and fold the norm of an un-reduced gradient)."""
        self._bwd_hooks = hooks or {}
        if hooks and not self._data_parallel:
            self._data_parallel = True
            self._bwd_plan = None
            self._bwd_plan_serial = None

    @property
    def _fold_sumsq_ok(self) -> bool:
        return self.first_touch and not self._data_parallel

    # ------------------------------------------------------------------------------------ parameters
    def param_tree(self) -> dict:
        return self.layout.flax_tree(self.params)

    def grad_tree(self) -> dict:
        return self.layout.flax_tree(self.grads)

    def load_params(self, tree: dict):
        """Copy a Flax-shaped tree (numpy arrays or tensors) into the flat buffer."""
        src = tree["params"] if "params" in tree else tree
        _copy_tree(self.param_tree()["params"], src)
        self.weights_stale = True

    def init_params(self, seed: int = 0):
        """Reference initialisers: lecun-normal Dense kernels (truncated at 2 sigma), zero biases,
        zero cls (vit.py:83), normal(0.02) pos-embed (position_embed.py:49), LN scale 1 / bias 0, ZERO head kernel
        (vit.py:98)."""
        g = torch.Generator(device="cpu").manual_seed(int(seed))
        self.params.zero_()
        lay, cfg = self.layout, self.cfg

        def lecun(name, fan_in):
            o, shape = lay.off[name]
            std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
            t = torch.empty(shape, dtype=f32)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=g)
            lay.view(self.params, name).copy_(t)

        lecun("Wpe", cfg.patch_dim)
        lay.view(self.params, "pos").copy_(torch.randn(lay.off["pos"][1], generator=g) * 0.02)
        for l in range(cfg.num_layers):
            lay.view(self.params, f"l{l}.ln1_g").fill_(1.0)
            lay.view(self.params, f"l{l}.ln2_g").fill_(1.0)
            lecun(f"l{l}.Wqkv", cfg.embed_dim)
            lecun(f"l{l}.Wo", cfg.embed_dim)
            lecun(f"l{l}.W1", cfg.embed_dim)
            lecun(f"l{l}.W2", cfg.hidden)
        lay.view(self.params, "lnf_g").fill_(1.0)
        self.weights_stale = True

    # ------------------------------------------------------------------------------------ plans
    def _off_ptr(self, buf: torch.Tensor, name: str) -> int:
        return buf.data_ptr() + self.layout.off[name][0] * 4

    def _gemm(self, plan: _Plan, label: str, writes: tuple = (), **kw):
        a = _lib.GemmArgs()
        for k, v in kw.items():
            setattr(a, k, v)
        if not a.rows_per_sample:
            a.rows_per_sample = 1
        a.round_bias_bf16 = self.rp
        a.cu_budget = self.cu_budget if (self.reserved_cus and self._building_bwd) else 0  # the all-reduce is resident during backward only
        if not a.tile and a.M <= 256 and not self.opt.rows_tile:
            # A/B runs (EngineOptions.rows_tile=False): few-row products on the LDS tile a many-row product of the same width takes instead
            # of the few-rows kernel (tile 24).  Both walk K in the same order per output element: bit-identical results, other launch shape
            a.tile = int(self.L.savit_gemm_tn_auto_tile_cus(257, a.N, a.K, a.epilogue, a.cu_budget))
        plan.keep.append(a)
        plan.add(self.L.savit_gemm_bf16_tn, (ctypes.byref(a),), label, writes=writes)

    def _build_cast_plan(self) -> _Plan:
        P, L, lay, cfg = _Plan(), self.L, self.layout, self.cfg
        d, F, C, NL = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.num_layers
        ls = lay.layer_stride
        mp = lambda n: self.params_bf16.data_ptr() + lay.off[n][0] * 2  # noqa: E731
        # the four weight families of every layer and the patch embedding in ONE launch (round 5; five launches before)
        jobs = (_lib.TransposeJob * 5)()
        for q, (name, R, Cc) in zip(jobs, (("Wqkv", d, 3 * d), ("Wo", d, d), ("W1", d, F), ("W2", F, d))):
            q.src, q.dst, q.src_batch_stride, q.dst_batch_stride = mp(f"l0.{name}"), self.w[name + "_t"].data_ptr(), ls, R * Cc
            q.ld_src, q.ld_dst, q.batch, q.rows, q.cols = Cc, R, NL, R, Cc
        q = jobs[4]
        q.src, q.dst, q.src_batch_stride, q.dst_batch_stride = mp("Wpe"), self.w["Wpe_t"].data_ptr(), 0, 0
        q.ld_src, q.ld_dst, q.batch, q.rows, q.cols = d, cfg.patch_dim, 1, cfg.patch_dim, d
        P.keep.append(jobs)
        P.add(L.savit_transpose_bf16_jobs, (jobs, 5), "cast weights")
        P.add(L.savit_cast_transpose_bf16, (self._off_ptr(self.params, "Wh"), 0, 1, d, C, self.w["Wh_n"].data_ptr(), 0, self.Cp,
                                            self.w["Wh_t"].data_ptr(), 0, d), "cast Wh")
        return P

    def _build_fwd_plan(self) -> _Plan:
        P, L, cfg = _Plan(), self.L, self.cfg
        d, F, C, N, NL, H, B, M = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers, cfg.num_heads, self.B, self.M
        pp = lambda n: self._off_ptr(self.params, n)  # noqa: E731
        x = self.x
        # tokens: patch-embed GEMM writes rows 1.. of each image, cls kernel writes row 0  (vit.py:77-85, position_embed.py:56)
        self._gemm(P, "patch_embed", A=self._img_buf.data_ptr(), Bt=self.w["Wpe_t"].data_ptr(), C=x[0].data_ptr(), aux=pp("pos"),
                   M=B * cfg.n_patches, N=d, K=cfg.patch_dim, lda=0, ldb=cfg.patch_dim, ldc=d, ldaux=d, epilogue=_lib.EPI_PATCH,
                   img_size=cfg.img_size, patch=cfg.patch, tokens=N, token_offset=1)
        P.add(L.savit_cls_pos_rows, (pp("cls"), pp("pos"), x[0].data_ptr(), B, N * d, d), "cls_rows")
        alpha = 1.0 / math.sqrt(cfg.head_dim)
        cfw = bool(self.cls_fwd)
        cb = self._cls_buffers() if cfw else None
        for l in range(NL):
            st = self.stats[l]
            w = lambda n, l=l: self.w[n][l].data_ptr()  # noqa: E731
            P.add(L.savit_layernorm_fwd, (x[l].data_ptr(), pp(f"l{l}.ln1_g"), pp(f"l{l}.ln1_b"), self.h1[l].data_ptr(), st[0].data_ptr(),
                                          st[1].data_ptr(), M, d, d, 1e-6, self.rp), f"l{l}.ln1")
            self._gemm(P, f"l{l}.qkv", A=self.h1[l].data_ptr(), Bt=w("Wqkv_t"), C=self.qkv[l].data_ptr(), M=M, N=3 * d, K=d, lda=d, ldb=d,
                       ldc=3 * d, epilogue=_lib.EPI_BF16, alpha=alpha, alpha_cols=d)
            if cfw and l == NL - 1:
                # the last layer behind its qkv projection, cls rows only (B rows; row b of the dense tensors at pitch N * width): only row 0
                # of its output is ever read.  K and V of every token are in qkv (dense: the cls query attends to all of them).
                P.add(L.savit_cls_query_attention_fwd, (self.qkv[l].data_ptr(), N * 3 * d, self.qkv[l].data_ptr() + d * 2, 3 * d, cb["o"].data_ptr(),
                                                        cb["lse"].data_ptr(), B, N, H, cfg.head_dim), f"l{l}.attn")
                self._gemm(P, f"l{l}.proj", A=cb["o"].data_ptr(), Bt=w("Wo_t"), C=cb["xmid"].data_ptr(), aux=x[l].data_ptr(), M=B, N=d, K=d,
                           lda=d, ldb=d, ldc=d, ldaux=N * d, epilogue=_lib.EPI_RESID)
                P.add(L.savit_layernorm_fwd, (cb["xmid"].data_ptr(), pp(f"l{l}.ln2_g"), pp(f"l{l}.ln2_b"), cb["h2"].data_ptr(),
                                              cb["stats"][0].data_ptr(), cb["stats"][1].data_ptr(), B, d, d, 1e-6, self.rp), f"l{l}.ln2")
                self._gemm(P, f"l{l}.fc1", A=cb["h2"].data_ptr(), Bt=w("W1_t"), C=cb["u"].data_ptr(), C2=cb["a"].data_ptr(),
                           bias=pp(f"l{l}.b1"), M=B, N=F, K=d, lda=d, ldb=d, ldc=F, epilogue=_lib.EPI_BIAS_GELU)
                self._gemm(P, f"l{l}.fc2", A=cb["a"].data_ptr(), Bt=w("W2_t"), C=cb["xout"].data_ptr(), bias=pp(f"l{l}.b2"),
                           aux=cb["xmid"].data_ptr(), M=B, N=d, K=F, lda=F, ldb=F, ldc=d, ldaux=d, epilogue=_lib.EPI_RESID)
                continue
            P.add(L.savit_attention_fwd, (self.qkv[l].data_ptr(), self.o[l].data_ptr(), self.lse[l].data_ptr(), B, N, H, cfg.head_dim, 3 * d),
                  f"l{l}.attn")
            self._gemm(P, f"l{l}.proj", A=self.o[l].data_ptr(), Bt=w("Wo_t"), C=self.xmid[l].data_ptr(), aux=x[l].data_ptr(), M=M, N=d, K=d,
                       lda=d, ldb=d, ldc=d, ldaux=d, epilogue=_lib.EPI_RESID)
            P.add(L.savit_layernorm_fwd, (self.xmid[l].data_ptr(), pp(f"l{l}.ln2_g"), pp(f"l{l}.ln2_b"), self.h2[l].data_ptr(),
                                          st[2].data_ptr(), st[3].data_ptr(), M, d, d, 1e-6, self.rp), f"l{l}.ln2")
            self._gemm(P, f"l{l}.fc1", A=self.h2[l].data_ptr(), Bt=w("W1_t"), C=self.u[l].data_ptr(), C2=self.a[l].data_ptr(),
                       bias=pp(f"l{l}.b1"), M=M, N=F, K=d, lda=d, ldb=d, ldc=F, epilogue=_lib.EPI_BIAS_GELU)
            self._gemm(P, f"l{l}.fc2", A=self.a[l].data_ptr(), Bt=w("W2_t"), C=x[l + 1].data_ptr(), bias=pp(f"l{l}.b2"),
                       aux=self.xmid[l].data_ptr(), M=M, N=d, K=F, lda=F, ldb=F, ldc=d, ldaux=d, epilogue=_lib.EPI_RESID)
        # final LayerNorm on the cls rows only (vit.py:57,95: only row 0 reaches the head), then the head Dense
        xlast, xlast_stride = (cb["xout"].data_ptr(), d) if cfw else (x[NL].data_ptr(), N * d)
        P.add(L.savit_layernorm_fwd, (xlast, pp("lnf_g"), pp("lnf_b"), self.zcls.data_ptr(), self.fstats[0].data_ptr(),
                                      self.fstats[1].data_ptr(), B, d, xlast_stride, 1e-6, self.rp), "lnf")
        self._gemm(P, "head", A=self.zcls.data_ptr(), Bt=self.w["Wh_t"].data_ptr(), C=self.logits.data_ptr(), bias=pp("bh"), M=B, N=C, K=d,
                   lda=d, ldb=d, ldc=C, epilogue=_lib.EPI_F32, round_out_bf16=self.rp)
        return P

    def _build_bwd_plan(self) -> _Plan:
        self._building_bwd = True
        try:
            return self._record_bwd_plan()
        finally:
            self._building_bwd = False

    def _record_bwd_plan(self) -> _Plan:
        P, L, cfg = _Plan(), self.L, self.cfg
        d, F, C, N, NL, H, B, M = cfg.embed_dim, cfg.hidden, cfg.num_classes, cfg.seq_len, cfg.num_layers, cfg.num_heads, self.B, self.M
        pp = lambda n: self._off_ptr(self.params, n)  # noqa: E731
        gp = lambda n: self._off_ptr(self.grads, n)  # noqa: E731
        ws, wsb = self.ln_ws.data_ptr(), self.ln_ws.numel()

        queue = WgradQueue(self.wgrad_cap, self.wgrad_max_lag) if self.wgrad_tile else None
        n_launch = [0]

        def wgrad(label, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, patch=(0, 0, 0, 0), layer=None, few_rows=False):
            # no later launch consumes dW: side stream.  X is a saved activation (stable until the next forward), dY is scratch
            if queue is not None and layer is not None and not (layer in self.wgrad_divert and label.endswith(".Wo.wgrad")):
                queue.push((X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw), layer, int(L.savit_gemm_wgrad_group_tiles(Kin, Nout, self.wgrad_tile)))
                return
            if few_rows and queue is not None and small_groups and Kin % 8 == 0 and Nout % 8 == 0:
                # a B-row product (the head; a cls-only weight that found no free slot in the tile FIFO): one workgroup per output tile
                # stores it by first touch - no K-split over 128 rows, no partial slabs, no reduce launch (round 5: 15 + 11 us -> one launch)
                tiles = int(L.savit_gemm_wgrad_group_tiles(Kin, Nout, self.wgrad_tile))
                if 0 < tiles <= self.wgrad_cap:
                    add_wgrad_group(self, P, label, [((X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw), 0, tiles)], self.wgrad_tile, [])
                    return
            self._add_wgrad(P, label, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, self._wgrad_splits(Kin, Nout, patch[0]), patch)

        wpe_g = bool(getattr(self, "_wpe_grouped", False)) and queue is not None and not self._data_parallel
        # (opt-in: W1 of the cls-only layer 15 + 11 us -> 14 us, but a 36-tile launch then sits among the 256-tile launches of the kernel
        #  the bench reports - its average launch time and rocprofv3's AverageNs would mix two kinds of launches for 12 us per step)
        small_groups = self.opt.wgrad_small_groups

        def flush_group(layer: int, last: bool, final: bool = False):
            # called between a layer's last input-gradient GEMM and its ln1.bwd (which overwrites the oldest ring slot): every
            # cotangent of the last `wgrad_lag` + 1 layers is still intact.  A launch takes one tile per CU; the DDP trigger
            # ('l{j}.ln1.bwd') of an EARLIER layer whose last tile is in it fires behind it, the current layer's own follows naturally.
            while queue is not None and queue.pending() > 0 and (queue.due(layer) or last):
                entries, done, oldest = queue.take(queue.cap)
                assert oldest - layer <= self.wgrad_lag, "weight-gradient queue reaches back further than the cotangent rings"
                add_wgrad_group(self, P, f"wgrad.group.{n_launch[0]}.l{oldest}-l{layer}", entries, self.wgrad_tile,
                                [f"l{j}.ln1.bwd" for j in done if (j != layer or final)] + (["Wpe.wgrad"] if final else []))
                n_launch[0] += 1

        # LayerNorm backward launches.  Alone on the GPU (no data-parallel bucket trigger needs the bias / LayerNorm gradients layer by
        # layer) every launch leaves its column-sum slab in a workspace of its own and ONE savit_layernorm_bwd_finalize_jobs launch at
        # the end of backward reduces them all (25 finalize launches of 7.5 us + a kernel boundary each -> 1); otherwise each call
        # finalizes itself as before.
        defer = self.defer_ln_finalize and not self._data_parallel and d > 64
        jobs: List[tuple] = []

        def ln_bwd(label, head8, outs3, tail5, extra=None, writes=(), sparse=None):
            rows = tail5[0]
            if sparse is not None:
                # savit_layernorm_bwd_sparse: statistics at a row stride, residual gradient of every res_mod-th row only (compact)
                stat_stride, res_mod, res_stride = sparse
                wbuf, outs = (ws, wsb), outs3
                if defer:
                    wb = self._ln_ws_slot(len(jobs), rows, d)
                    wbuf, outs = (wb.data_ptr(), wb.numel()), (None, None, None)
                ex = extra if (extra is not None and not defer) else (None, 0, 0, None)
                P.add(L.savit_layernorm_bwd_sparse, head8[:5] + (stat_stride, head8[5], res_mod, res_stride) + head8[6:8] + outs + tail5 + wbuf + ex,
                      label, writes=writes)
                if defer:
                    jobs.append((wbuf[0], int(L.savit_layernorm_bwd_grid(rows)), d, 3, outs3 + (None,), extra))
                return
            if defer:
                k = len(jobs)
                wbuf = self._ln_ws_slot(k, rows, d)
                P.add(L.savit_layernorm_bwd, head8 + (None, None, None) + tail5 + (wbuf.data_ptr(), wbuf.numel()), label, writes=writes)
                jobs.append((wbuf.data_ptr(), int(L.savit_layernorm_bwd_grid(rows)), d, 3, outs3 + (None,), extra))
            elif extra is not None:
                P.add(L.savit_layernorm_bwd_ex, head8 + outs3 + tail5 + (ws, wsb) + extra, label, writes=writes)
            else:
                P.add(L.savit_layernorm_bwd, head8 + outs3 + tail5 + (ws, wsb), label, writes=writes)

        def flush_ln_jobs(label):
            if not jobs:
                return
            arr = (_lib.ColsumJob * len(jobs))()
            for q, (partial, nblk, dd, nf, outs, extra) in zip(arr, jobs):
                q.partial, q.nblk, q.d, q.nf = partial, nblk, dd, nf
                for i, o in enumerate(outs):
                    q.out[i] = o
                if extra is not None:
                    q.extra_slab, q.extra_rows, q.extra_n, q.extra_out = extra
            P.keep.append(arr)
            P.add(L.savit_layernorm_bwd_finalize_jobs, (arr, len(jobs)), label)
            del jobs[:]

        ring, ri = [t.data_ptr() for t in self.dres_b_ring], 0
        # ---- head: dWh, d z_cls, final LayerNorm backward into the (zeroed) residual gradient
        wgrad("head.wgrad", self.zcls.data_ptr(), self.dlogits.data_ptr(), gp("Wh"), B, d, C, d, self.Cp, C)  # (as one grouped launch: 16 us against 8 + 5 for the split pair)
        self._gemm(P, "head.dgrad", A=self.dlogits.data_ptr(), Bt=self.w["Wh_n"].data_ptr(), C=self.d_z.data_ptr(), M=B, N=d, K=self.Cp,
                   lda=self.Cp, ldb=self.Cp, ldc=d, epilogue=_lib.EPI_BF16)
        cls_last = bool(self.cls_only_last)
        self._needs_zero_dres = not cls_last
        cb = self._cls_buffers() if cls_last else None
        cfw = cls_last and bool(self.cls_fwd)
        if cls_last:
            # compact [B, d] residual gradient of the cls rows (fp32) and its bf16 copy: no [B*N, d] buffer is zero-filled to carry B rows
            xl, xls = (cb["xout"].data_ptr(), d) if cfw else (self.x[NL].data_ptr(), N * d)
            ln_bwd("lnf.bwd", (self.d_z.data_ptr(), xl, pp("lnf_g"), self.fstats[0].data_ptr(), self.fstats[1].data_ptr(), None,
                               cb["dres"].data_ptr(), cb["rb0"].data_ptr()), (gp("lnf_g"), gp("lnf_b"), gp(f"l{NL - 1}.b2")), (B, d, xls, d, self.rp))
        else:
            ln_bwd("lnf.bwd", (self.d_z.data_ptr(), self.x[NL].data_ptr(), pp("lnf_g"), self.fstats[0].data_ptr(), self.fstats[1].data_ptr(), None,
                               self.dres.data_ptr(), ring[0]), (gp("lnf_g"), gp("lnf_b"), gp(f"l{NL - 1}.b2")), (B, d, N * d, N * d, self.rp),
                   writes=(ring[0],))
        for l in range(NL - 1, -1, -1):
            st = self.stats[l]
            w = lambda n, l=l: self.w[n][l].data_ptr()  # noqa: E731
            d_u, dqkv = self.d_u_ring[l % len(self.d_u_ring)].data_ptr(), self.dqkv_ring[l % len(self.dqkv_ring)].data_ptr()
            if cls_last and l == NL - 1:
                # ---- the LAST layer, cls rows only (B rows at pitch N * width inside the saved activations; compact cotangents).  Same
                # kernels, same arithmetic per row: the rows left out contribute exact zeros to every sum below.
                rb0, rb1, du_c, dh_c, dres_c, cs = (cb[k].data_ptr() for k in ("rb0", "rb1", "d_u", "d_h", "dres", "slab"))
                if cfw:   # compact activations (the forward ran on the cls rows: _build_fwd_plan)
                    a_p, a_ld, u_p, u_ld, h2_p, h2_ld = cb["a"].data_ptr(), F, cb["u"].data_ptr(), F, cb["h2"].data_ptr(), d
                    xm_p, xm_ld, m2, r2, sstr, o_p, o_ld = cb["xmid"].data_ptr(), d, cb["stats"][0].data_ptr(), cb["stats"][1].data_ptr(), 1, cb["o"].data_ptr(), d
                else:     # the cls rows of the dense activations: row b at pitch N * width
                    a_p, a_ld, u_p, u_ld, h2_p, h2_ld = self.a[l].data_ptr(), N * F, self.u[l].data_ptr(), N * F, self.h2[l].data_ptr(), N * d
                    xm_p, xm_ld, m2, r2, sstr, o_p, o_ld = self.xmid[l].data_ptr(), N * d, st[2].data_ptr(), st[3].data_ptr(), N, self.o[l].data_ptr(), N * d
                cq = lambda n: (None if n in self._cls_per_weight else l)  # noqa: E731  (a tile of the grouped launches, or a launch of its own)
                wgrad(f"l{l}.W2.wgrad", a_p, rb0, gp(f"l{l}.W2"), B, F, d, a_ld, d, d, layer=cq("W2"), few_rows=True)
                self._gemm(P, f"l{l}.fc2.dgrad", A=rb0, Bt=w("W2_n"), C=du_c, aux=u_p, colsum=cs, colsum_rows=cb["slab"].shape[0],
                           M=B, N=F, K=d, lda=d, ldb=d, ldc=F, ldaux=u_ld, epilogue=_lib.EPI_DGELU)
                wgrad(f"l{l}.W1.wgrad", h2_p, du_c, gp(f"l{l}.W1"), B, d, F, h2_ld, F, F, layer=cq("W1"), few_rows=True)
                self._gemm(P, f"l{l}.fc1.dgrad", A=du_c, Bt=w("W1_n"), C=dh_c, M=B, N=d, K=F, lda=F, ldb=F, ldc=d, epilogue=_lib.EPI_BF16)
                ln_bwd(f"l{l}.ln2.bwd", (dh_c, xm_p, pp(f"l{l}.ln2_g"), m2, r2, dres_c, dres_c, rb1),
                       (gp(f"l{l}.ln2_g"), gp(f"l{l}.ln2_b"), None), (B, d, xm_ld, d, self.rp),
                       extra=(cs, cb["slab"].shape[0], F, gp(f"l{l}.b1")), sparse=(sstr, 0, 0))
                wgrad(f"l{l}.Wo.wgrad", o_p, rb1, gp(f"l{l}.Wo"), B, d, d, o_ld, d, d, layer=cq("Wo"), few_rows=True)
                if cfw:
                    # the cls query's attention backward: dQ at the cls rows, dK / dV of every key; the q columns of the other rows of this
                    # layer's own cotangent buffer are zero and stay zero
                    dqkv = cb["dqkv"].data_ptr()
                    self._gemm(P, f"l{l}.proj.dgrad", A=rb1, Bt=w("Wo_n"), C=cb["d_o"].data_ptr(), M=B, N=d, K=d, lda=d, ldb=d, ldc=d,
                               epilogue=_lib.EPI_BF16)
                    P.add(L.savit_cls_query_attention_bwd, (self.qkv[l].data_ptr(), N * 3 * d, self.qkv[l].data_ptr() + d * 2, 3 * d, o_p,
                                                            cb["lse"].data_ptr(), cb["d_o"].data_ptr(), dqkv, N * 3 * d, dqkv + d * 2, B, N, H,
                                                            cfg.head_dim, 1.0 / math.sqrt(cfg.head_dim)), f"l{l}.attn.bwd", writes=(dqkv,))
                else:
                    # attention backward reads every row of d_o: zero it, then the projection's input gradient fills the cls rows
                    P.add(L.savit_zero_bytes, (self.d_o.data_ptr(), self.d_o.numel() * 2), "zero.d_o")
                    self._gemm(P, f"l{l}.proj.dgrad", A=rb1, Bt=w("Wo_n"), C=self.d_o.data_ptr(), M=B, N=d, K=d, lda=d, ldb=d, ldc=N * d,
                               epilogue=_lib.EPI_BF16)
                    P.add(L.savit_attention_bwd, (self.qkv[l].data_ptr(), self.o[l].data_ptr(), self.d_o.data_ptr(), self.lse[l].data_ptr(),
                                                  dqkv, B, N, H, cfg.head_dim, 3 * d, 1.0 / math.sqrt(cfg.head_dim)), f"l{l}.attn.bwd", writes=(dqkv,))
                wgrad(f"l{l}.Wqkv.wgrad", self.h1[l].data_ptr(), dqkv, gp(f"l{l}.Wqkv"), M, d, 3 * d, d, 3 * d, 3 * d, layer=l)
                self._gemm(P, f"l{l}.qkv.dgrad", A=dqkv, Bt=w("Wqkv_n"), C=self.d_h.data_ptr(), M=M, N=d, K=3 * d, lda=3 * d, ldb=3 * d, ldc=d,
                           epilogue=_lib.EPI_BF16)
                flush_group(l, last=(l == 0 and not wpe_g))
                ri = (ri + 1) % len(ring)
                # the first dense LayerNorm backward merges the compact residual gradient of the cls rows (rows r % N == 0)
                ln_bwd(f"l{l}.ln1.bwd", (self.d_h.data_ptr(), self.x[l].data_ptr(), pp(f"l{l}.ln1_g"), st[0].data_ptr(), st[1].data_ptr(), dres_c,
                                         self.dres.data_ptr(), ring[ri]),
                       (gp(f"l{l}.ln1_g"), gp(f"l{l}.ln1_b"), gp(f"l{l - 1}.b2") if l > 0 else None), (M, d, d, d, self.rp), writes=(ring[ri],),
                       sparse=(1, N, d))
                continue
            # FFN branch: x_{l+1} = x_mid + gelu(h2 W1 + b1) W2 + b2     (ff.py:26-33, vit.py:26-31)
            wgrad(f"l{l}.W2.wgrad", self.a[l].data_ptr(), ring[ri], gp(f"l{l}.W2"), M, F, d, F, d, d, layer=l)
            # db1 = column sums of d_u: per-row-tile partials (plain stores) + a finalize launch; ~200 row tiles adding into the
            # same F addresses with atomics serialise at the memory side (16 us of this 180 us launch), and this is reproducible
            cslab = self._colsum_slab_for(l) if defer else self.colsum_slab  # (deferred: the slab lives until the end of backward)
            self._gemm(P, f"l{l}.fc2.dgrad", writes=(d_u,), A=ring[ri], Bt=w("W2_n"), C=d_u, aux=self.u[l].data_ptr(),
                       colsum=cslab.data_ptr(), colsum_rows=cslab.shape[0], M=M, N=F, K=d, lda=d, ldb=d, ldc=F,
                       ldaux=F, epilogue=_lib.EPI_DGELU)
            # the slab is reduced into db1 by the finalize launch of this layer's ln2.bwd below (savit_layernorm_bwd_ex: wide rows only)
            if d <= 64:
                P.add(L.savit_colsum_finalize, (self.colsum_slab.data_ptr(), self.colsum_slab.shape[0], F, gp(f"l{l}.b1"), 1), f"l{l}.b1.grad")
            wgrad(f"l{l}.W1.wgrad", self.h2[l].data_ptr(), d_u, gp(f"l{l}.W1"), M, d, F, d, F, F, layer=l)
            self._gemm(P, f"l{l}.fc1.dgrad", A=d_u, Bt=w("W1_n"), C=self.d_h.data_ptr(), M=M, N=d, K=F, lda=F, ldb=F,
                       ldc=d, epilogue=_lib.EPI_BF16)
            ri = (ri + 1) % len(ring)
            ln_bwd(f"l{l}.ln2.bwd", (self.d_h.data_ptr(), self.xmid[l].data_ptr(), pp(f"l{l}.ln2_g"), st[2].data_ptr(), st[3].data_ptr(),
                                     self.dres.data_ptr(), self.dres.data_ptr(), ring[ri]), (gp(f"l{l}.ln2_g"), gp(f"l{l}.ln2_b"), None),
                   (M, d, d, d, self.rp), extra=(cslab.data_ptr(), cslab.shape[0], F, gp(f"l{l}.b1")) if d > 64 else None, writes=(ring[ri],))
            # attention branch: x_mid = x_l + attn(LN1(x_l)) Wo     (attention.py:21-67, vit.py:19-24)
            wgrad(f"l{l}.Wo.wgrad", self.o[l].data_ptr(), ring[ri], gp(f"l{l}.Wo"), M, d, d, d, d, d, layer=l)
            self._gemm(P, f"l{l}.proj.dgrad", A=ring[ri], Bt=w("Wo_n"), C=self.d_o.data_ptr(), M=M, N=d, K=d, lda=d, ldb=d,
                       ldc=d, epilogue=_lib.EPI_BF16)
            P.add(L.savit_attention_bwd, (self.qkv[l].data_ptr(), self.o[l].data_ptr(), self.d_o.data_ptr(), self.lse[l].data_ptr(),
                                          dqkv, B, N, H, cfg.head_dim, 3 * d, 1.0 / math.sqrt(cfg.head_dim)), f"l{l}.attn.bwd",
                  writes=(dqkv,))
            wgrad(f"l{l}.Wqkv.wgrad", self.h1[l].data_ptr(), dqkv, gp(f"l{l}.Wqkv"), M, d, 3 * d, d, 3 * d, 3 * d, layer=l)
            self._gemm(P, f"l{l}.qkv.dgrad", A=dqkv, Bt=w("Wqkv_n"), C=self.d_h.data_ptr(), M=M, N=d, K=3 * d, lda=3 * d,
                       ldb=3 * d, ldc=d, epilogue=_lib.EPI_BF16)
            flush_group(l, last=(l == 0 and not wpe_g))
            ri = (ri + 1) % len(ring)
            ln_bwd(f"l{l}.ln1.bwd", (self.d_h.data_ptr(), self.x[l].data_ptr(), pp(f"l{l}.ln1_g"), st[0].data_ptr(), st[1].data_ptr(),
                                     self.dres.data_ptr(), self.dres.data_ptr(), ring[ri]),
                   (gp(f"l{l}.ln1_g"), gp(f"l{l}.ln1_b"), gp(f"l{l - 1}.b2") if l > 0 else None), (M, d, d, d, self.rp), writes=(ring[ri],))
        # ---- embeddings: dpos, dcls, dWpe   (vit.py:77-85, position_embed.py:56, patch_embed.py:23-25)
        P.add(L.savit_pos_cls_grad, (self.dres.data_ptr(), gp("pos"), gp("cls"), B, N, d, 1), "pos_cls.grad")
        if wpe_g:
            # the patch-embed weight gradient as tiles of the last grouped launch: X = the dense patch matrix, one row per TOKEN (the cls
            # rows stay zero), so that it lines up with the bf16 residual gradient row by row
            pm = self._patch_matrix()
            P.add(L.savit_patchify_bf16, (self._img_buf.data_ptr(), pm.data_ptr(), B, cfg.img_size, cfg.patch, N, 1), "patchify")
            queue.push((pm.data_ptr(), ring[ri], gp("Wpe"), M, cfg.patch_dim, d, cfg.patch_dim, d, d), 0,
                       int(L.savit_gemm_wgrad_group_tiles(cfg.patch_dim, d, self.wgrad_tile)))
            flush_group(0, last=True, final=True)
        else:
            wgrad("Wpe.wgrad", self._img_buf.data_ptr(), ring[ri], gp("Wpe"), B * cfg.n_patches, cfg.patch_dim, d, 0, d, d,
                  patch=(cfg.patch, cfg.img_size, N, 1))
        flush_ln_jobs("ln.bwd.finalize")
        finalize_wgrad_ws(self, P)
        finalize_first_touch(self, P)
        return P

    def _ln_ws_slot(self, k: int, rows: int, d: int) -> torch.Tensor:
        """Workspace number k of the deferred LayerNorm-backward column sums (one per launch of a backward pass; shared by the plans)."""
        if not hasattr(self, "_ln_ws_slots"):
            self._ln_ws_slots = []
        need = max(16, int(self.L.savit_layernorm_bwd_workspace_bytes(rows, d)))
        while len(self._ln_ws_slots) <= k:
            self._ln_ws_slots.append(None)
        buf = self._ln_ws_slots[k]
        if buf is None or buf.numel() < need:
            buf = self._ln_ws_slots[k] = torch.empty(need, dtype=torch.uint8, device=self.dev)
        return buf

    def _cls_buffers(self) -> dict:
        """Compact cotangents of the cls rows for the last layer's backward (cls_only_last): [B, d] / [B, F], and the bias-gradient slab
        of its B-row GELU' GEMM."""
        if not hasattr(self, "_cls_bufs"):
            cfg, B = self.cfg, self.B
            d, F = cfg.embed_dim, cfg.hidden
            e = lambda *s_, dt=bf16: torch.empty(*s_, dtype=dt, device=self.dev)  # noqa: E731
            rows = max(1, int(self.L.savit_gemm_colsum_rows_cus(B, F, d, 0, self.cu_budget if self.reserved_cus else 0)))
            self._cls_bufs = {"dres": e(B, d, dt=f32), "rb0": e(B, d), "rb1": e(B, d), "d_u": e(B, F), "d_h": e(B, d), "slab": e(rows, F, dt=f32)}
            if self.cls_fwd:  # the last layer's activations behind its qkv projection, cls rows only
                H, M = cfg.num_heads, self.M
                self._cls_bufs.update({"o": e(B, d), "lse": e(B, H, dt=f32), "xmid": e(B, d, dt=f32), "h2": e(B, d), "stats": e(2, B, dt=f32),
                                       "u": e(B, F), "a": e(B, F), "xout": e(B, d, dt=f32), "d_o": e(B, d),
                                       # q columns are written at the cls rows only and stay zero elsewhere: never shared with other layers
                                       "dqkv": torch.zeros(M, 3 * d, dtype=bf16, device=self.dev)})
        return self._cls_bufs

    def _patch_matrix(self) -> torch.Tensor:
        """[B * N, patch_dim] bf16: the patches of the current images at their tokens' rows (savit_patchify_bf16); the cls rows stay zero."""
        if not hasattr(self, "_patches"):
            self._patches = torch.zeros(self.B * self.cfg.seq_len, self.cfg.patch_dim, dtype=bf16, device=self.dev)
        return self._patches

    def _colsum_slab_for(self, l: int) -> torch.Tensor:
        if not hasattr(self, "_colsum_slabs"):
            self._colsum_slabs = {}
        if l not in self._colsum_slabs:
            self._colsum_slabs[l] = torch.empty_like(self.colsum_slab)
        return self._colsum_slabs[l]

    def _add_wgrad(self, plan: "_Plan", label: str, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, splits: int, patch=(0, 0, 0, 0),
                   side: bool = True):
        add_wgrad(self, plan, label, X, dY, dW, Mr, Kin, Nout, ldx, lddy, lddw, splits, patch, side)

    def _wgrad_group_plan(self):
        """(tile, tiles per launch, layers whose Wo keeps the per-weight path, layers a launch reaches back); tile 0 = one launch per
        weight (token range split over workgroups, partial slabs + ordered reduce).
        A layer's four weight gradients are d x 3d, d x d, d x F, F x d (edge tiles where a side is not a multiple of 256: DeiT-S's
        d = 384 takes 38 tiles per layer, 71 % of them inside a matrix): with 256 x 256 tiles DeiT-B has 108 tiles per layer, ViT-L 192 -
        too few for 256 CUs one weight (or one layer) at a time.  They wait in a FIFO and leave in launches of exactly one tile per CU,
        a weight cut between two launches where needed: every launch is a full round.  DeiT-B: 12 x 108 = 1 296 tiles = 5 rounds + 16
        tiles - so the d x d gradients of the last two layers (18 tiles) keep the per-weight path and the rest is 5 launches (4 x 256 +
        254); ViT-L: 24 x 192 = 18 x 256 exactly.  SAVIT_WGRAD_GROUP=0 turns the grouping off."""
        cfg = self.cfg
        d, F, NL = cfg.embed_dim, cfg.hidden, cfg.num_layers
        if not self.opt.wgrad_group or d % 8 or F % 8:
            return 0, 0, frozenset(), 0
        tile = wgrad_group_tile(d, F, self.opt.wgrad_tile)
        sizes = [(n, int(self.L.savit_gemm_wgrad_group_tiles(a, b, tile))) for n, a, b in (("W2", F, d), ("W1", d, F), ("Wo", d, d), ("Wqkv", d, 3 * d))]
        per_layer = sum(t for _, t in sizes)
        cap = self.cu_budget
        cls_last = bool(getattr(self, "cls_only_last", False)) and type(self)._record_bwd_plan is ViTEngine._record_bwd_plan
        # The last layer's W2 / W1 / Wo gradients are 128-row products (cls rows only).  As tiles of the grouped launches they cost
        # nothing where the last round has free slots (their tiles end after 4 stages of tokens); the ones that do not fit keep a launch
        # of their own.  Their cotangents live in compact buffers nothing overwrites during backward, so the reach-back does not matter.
        skip_last = {"W2", "W1", "Wo"} if cls_last else set()
        total = NL * per_layer - sum(t for n, t in sizes if n in skip_last)
        if cls_last:
            free = -(-total // cap) * cap - total
            for n in ("W2", "W1", "Wo"):
                t = dict(sizes)[n]
                if t <= free:
                    skip_last.discard(n)
                    free -= t
                    total += t
        self._cls_per_weight = frozenset(skip_last)
        # the patch-embed weight gradient (patch_embed.py:23-25) as tiles of the LAST grouped launch, when that round has free slots: its X
        # operand is then the dense patch matrix savit_patchify_bf16 writes (20 us) instead of a gather inside a launch of its own (84 + 6 us)
        wpe_tiles = int(self.L.savit_gemm_wgrad_group_tiles(cfg.patch_dim, d, tile)) if cfg.patch % 8 == 0 and cfg.patch_dim % 8 == 0 else 0
        free = -(-total // cap) * cap - total
        self._wpe_grouped = bool(self.opt.wpe_grouped and type(self)._record_bwd_plan is ViTEngine._record_bwd_plan
                                 and 0 < wpe_tiles <= free and total > 0 and self.wgrad_max_lag is None)  # (a bound on the reach-back wins)
        if self._wpe_grouped:
            total += wpe_tiles
        rounds = -(-total // cap)
        need = total - (rounds - 1) * cap  # tiles in the last, partial round
        wo = dict(sizes)["Wo"]
        divert = frozenset()
        if rounds >= 2 and need <= 0.12 * cap and -(-need // wo) <= NL:
            divert = frozenset(range(-(-need // wo)))  # the layers processed LAST (0, 1, ...): their d x d gradients run per weight
        # dry run of the queue: how many layers does a launch reach back?
        q, lag = WgradQueue(cap, self.wgrad_max_lag), 0
        for l in range(NL - 1, -1, -1):
            for n, t in sizes:
                if not (n == "Wo" and l in divert) and not (l == NL - 1 and n in skip_last):
                    q.push(None, l, t)
            if l == 0 and getattr(self, "_wpe_grouped", False):
                q.push(None, 0, wpe_tiles)
            while q.pending() > 0 and (q.due(l) or l == 0):
                _, _, oldest = q.take(cap)
                lag = max(lag, oldest - l)
        if getattr(self, "_wpe_grouped", False):
            lag += 1  # the last launch goes out BEHIND layer 0's LayerNorm backward (it carries the patch-embed gradient): one more ring write
        return tile, cap, divert, lag

    def _wgrad_splits(self, Kin: int, Nout: int, patch: int) -> int:
        """K-splits of a weight-gradient GEMM.  On its own a launch wants every CU (0 = the library's choice); beside the
        main chain it is sized for `wgrad_cu_share` of them: measured on DeiT-B/16 the backward pass takes 14.2 ms with
        256-workgroup weight-gradient launches (7 splits) and 11.9 ms with 144-workgroup ones (4 splits, 43 % fewer atomics)."""
        if not self.overlap_wgrad or self._building_serial:
            return 0
        if self.L.savit_gemm_wgrad_auto_variant(Kin, Nout, patch) != 3:
            return 0  # small weights (128x128 tiles): the whole-chip launch is faster (DeiT-S: 17.3 k vs 16.3 k images/s)
        tiles = -(-Kin // 256) * -(-Nout // 256)
        return max(1, min(24, round(self.wgrad_cu_share * self.n_cus / tiles)))

    def _serial_bwd_plan(self) -> _Plan:
        if self._bwd_plan_serial is None:
            self._building_serial = True
            try:
                self._bwd_plan_serial = self._build_bwd_plan()
            finally:
                self._building_serial = False
        return self._bwd_plan_serial

    # ------------------------------------------------------------------------------------ execution
    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def refresh_weights(self):
        """fp32 master -> bf16 MFMA operands (both layouts).  Call after the parameters change."""
        if self._cast_plan is None:
            self._cast_plan = self._build_cast_plan()
        if getattr(self, "params_bf16", None) is not None and not self._mirror_fresh:
            # parameters changed outside an optimizer step (init, checkpoint load): rebuild the bf16 mirror from the fp32 master
            timed_call(self.launch_timer, "cast mirror", self.L.savit_cast_bf16, self.params.data_ptr(), self.params_bf16.data_ptr(),
                       self.params.numel(), self._stream())
        self._mirror_fresh = False
        self._cast_plan.run(self._stream(), self.launch_timer)
        self.weights_stale = False

    def set_images(self, images: torch.Tensor):
        """images: [B,S,S,3] NHWC (bf16 or fp32; train.py:81 casts to bf16) or the loader's [S,S,3,B] fp32 layout
        (train.py:80).  Copied into the engine's bf16 NHWC input buffer."""
        cfg = self.cfg
        if not images.is_cuda:
            raise ValueError("images must be on the GPU")
        S = cfg.img_size
        if tuple(images.shape) == (self.B, S, S, 3):
            if images.dtype == bf16:
                self._img_buf.copy_(images)
            elif images.dtype == f32:
                _lib.check(self.L.savit_cast_bf16(images.contiguous().data_ptr(), self._img_buf.data_ptr(), images.numel(), self._stream()),
                           "savit_cast_bf16")
            else:
                raise ValueError("images must be bf16 or fp32")
        elif tuple(images.shape) == (S, S, 3, self.B) and images.dtype == f32:
            _lib.check(self.L.savit_hwcn_to_nhwc_bf16(images.contiguous().data_ptr(), self._img_buf.data_ptr(), S, S, 3, self.B,
                                                      self._stream()), "savit_hwcn_to_nhwc_bf16")
        else:
            raise ValueError(f"images shape {tuple(images.shape)} does not match batch {self.B} / img_size {S}")

    def forward(self, images: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Runs the forward launch sequence; returns the engine-owned fp32 logits [B, C]."""
        if images is not None:
            self.set_images(images)
        if self.weights_stale:
            self.refresh_weights()
        if self._fwd_plan is None:
            self._fwd_plan = self._build_fwd_plan()
        self._fwd_plan.run(self._stream(), self.launch_timer)
        return self.logits

    def loss_backward(self, labels: torch.Tensor, label_smoothing: float = 0.1, mix_labels: Optional[torch.Tensor] = None,
                      ratio: Optional[torch.Tensor] = None, zero_grads: bool = True) -> torch.Tensor:
        """Loss (train.py:83-90) + full backward into self.grads.  Returns the device scalar loss."""
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
                   float(label_smoothing), 1.0 / self.B, self.loss_rows.data_ptr(), self.loss.data_ptr(),
                   self.dlogits.data_ptr(), self.Cp, self._off_ptr(self.grads, "bh"), self.top1.data_ptr(),
                   self.top5.data_ptr(), self.B, self.cfg.num_classes, s)
        self._backward_from_dlogits()
        return self.loss

    def _current_bwd_plan(self) -> _Plan:
        if self.overlap_wgrad:
            if self._bwd_plan is None:
                self._bwd_plan = self._build_bwd_plan()
            return self._bwd_plan
        return self._serial_bwd_plan()

    def _zero_grads_for_backward(self, zero_grads: bool):
        """Clear what backward ACCUMULATES into.  The matrices the grouped weight-gradient launches store by first touch are not
        cleared (savit_zero_ranges over the rest: biases, LayerNorm parameters, embeddings, head - 6 of DeiT-B's 346 MB); zero_grads =
        False (gradient accumulation over several backward passes) turns first touch off for this pass."""
        plan = self._current_bwd_plan()
        self._accumulate_run = not zero_grads
        if plan.fold_sumsq:
            self._zero("zero.gnorm", self.gnorm)
        if not zero_grads:
            return
        if plan.rest_ranges is not None:
            timed_call(self.launch_timer, "zero.grads", self.L.savit_zero_ranges, self.grads.data_ptr(), plan.rest_ranges, plan.n_rest, self._stream())
        else:
            self._zero("zero.grads", self.grads)

    def _zero(self, label: str, t: torch.Tensor):
        """hipMemsetAsync of an accumulator on the launch stream, as a labelled (timeable) launch."""
        timed_call(self.launch_timer, label, self.L.savit_zero_bytes, t.data_ptr(), t.numel() * t.element_size(), self._stream())

    def backward_from_dlogits(self, zero_grads: bool = True):
        """Custom-loss entry point: the caller has written self.dlogits (bf16 [B, Cp], pad columns zero).  The gradient buffer is
        prepared exactly as `loss_backward` prepares it - cleared where backward accumulates, untouched where the grouped launches store
        by first touch, the norm's fold slots zeroed; zero_grads=False keeps the buffer and makes every launch accumulate - and then the
        backward plan runs.  (this used to skip the preparation, so a direct caller got first-touch stores over a buffer it
        believed was accumulated into, and never-zeroed fold slots in the next clip norm.)  The head-bias gradient is written by
        savit_softmax_xent inside loss_backward: a custom loss adds its own column sum of dlogits into grad 'bh' after this call."""
        self._zero_grads_for_backward(zero_grads)
        self._backward_from_dlogits()

    def _backward_from_dlogits(self):
        """Backward from self.dlogits into self.grads; `_zero_grads_for_backward` has run (loss_backward / backward_from_dlogits)."""
        if getattr(self, "_needs_zero_dres", True):
            self._current_bwd_plan()  # (recording the plan decides whether the dense residual-gradient buffers carry the cls rows)
        if getattr(self, "_needs_zero_dres", True):
            self._zero("zero.dres", self.dres)
            self._zero("zero.dres_b", self.dres_b)  # ring slot 0: lnf.bwd fills only the cls rows
        self._run_bwd()

    def _run_bwd(self):
        """Issue the backward plan; the launch-time CU budget of the library (grids of the persistent attention kernels) is this
        engine's for the length of the issue and back to "every CU" afterwards."""
        if self.reserved_cus:
            _lib.check(self.L.savit_set_cu_budget(self.cu_budget), "savit_set_cu_budget")
        plan = ViTEngine._current_bwd_plan(self)  # (CaiTEngine borrows this method without deriving from ViTEngine)
        accumulate = getattr(self, "_accumulate_run", False)
        if accumulate:  # gradient accumulation: the grouped launches must add, and what they store is not the final gradient
            for arr in plan.wgrad_arrays:
                for q in arr:
                    q.overwrite = 0
        try:
            if self.overlap_wgrad:
                n = max(1, self.n_side_streams)
                while len(self._side_streams) < n:
                    self._side_streams.append(torch.cuda.Stream(device=self.dev))
                plan.run_overlapped(torch.cuda.current_stream(), self._side_streams[:n], self.bwd_hooks, self.launch_timer)
            else:
                plan.run(self._stream(), self.launch_timer, self.bwd_hooks)
            self._gnorm_folded = plan.fold_sumsq and not accumulate
            self._folded_plan = plan
        finally:
            if accumulate:
                g0 = self.grads.data_ptr()
                for arr in plan.wgrad_arrays:
                    for q in arr:
                        q.overwrite = 1 if (q.dW - g0) // 4 in plan.covered else 0
                self._accumulate_run = False
            if self.reserved_cus:
                _lib.check(self.L.savit_set_cu_budget(0), "savit_set_cu_budget")

    def optimizer_step(self, lr: float, weight_decay: float = 0.0, max_norm: float = 0.0, b1: float = 0.9, b2: float = 0.999,
                       eps: float = 1e-8, grad_scale: float = 1.0):
        """Fused AdamW over the flat buffers (train.py:25-27,100) + bf16 operand refresh."""
        if self.adam_m is None:
            self.adam_m = torch.zeros_like(self.params)
            self.adam_v = torch.zeros_like(self.params)
        s = self._stream()
        self.step_count += 1
        ss = None
        tm = self.launch_timer
        if max_norm and max_norm > 0:
            ss = self._grad_sumsq(tm, s)
        self._gnorm_folded = False
        mirror = getattr(self, "params_bf16", None)
        timed_call(tm, "adamw", self.L.savit_adamw_step_mirror, self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(),
                   self.adam_v.data_ptr(), self.params.numel(), float(lr), float(b1), float(b2), float(eps), float(weight_decay),
                   self.step_count, ss, float(max_norm or 0.0), float(grad_scale), mirror.data_ptr() if mirror is not None else None, s)
        self._mirror_fresh = mirror is not None
        self.refresh_weights()

    def _grad_sumsq(self, tm, s) -> int:
        """Squared global gradient norm into gnorm[0] (optax.clip_by_global_norm, train.py:25) -> its device address.  After a backward
        whose grouped launches accumulated the squares of the weight gradients they stored (gnorm[1:33]), only the ranges nothing
        overwrites are read (savit_sumsq_ranges); otherwise - data-parallel rank (the norm is the REDUCED gradient's), gradient
        accumulation, gradients written from outside - the whole buffer."""
        plan = getattr(self, "_folded_plan", None)
        if self._gnorm_folded and plan is not None and plan.rest_ranges is not None:
            timed_call(tm, "sumsq", self.L.savit_sumsq_ranges, self.grads.data_ptr(), plan.rest_ranges, plan.n_rest, self.gnorm.data_ptr() + 4, 32,
                       self.gnorm.data_ptr(), s)
        else:
            self._zero("zero.gnorm", self.gnorm_sq)
            timed_call(tm, "sumsq", self.L.savit_sumsq, self.grads.data_ptr(), self.grads.numel(), self.gnorm_sq.data_ptr(), s)
        return self.gnorm.data_ptr()

    def profile_step(self, labels: torch.Tensor, label_smoothing: float = 0.1, reps: int = 3) -> Dict[str, float]:
        """Forward + loss + backward with EVERY launch bracketed by timing events on the launch stream, each repetition enqueued
        behind a gate kernel (timing.instrumented_steps).  Returns {launch label: milliseconds, minimum over `reps`}."""
        from .timing import instrumented_steps

        def one():
            self.forward()
            self.loss_backward(labels, label_smoothing)

        return instrumented_steps(self, one, reps=reps)["labels"]

    def activation_bytes(self) -> int:
        tot = 0
        for group in (self.x, self.xmid, self.h1, self.h2, self.qkv, self.o, self.u, self.a, self.stats, self.lse):
            tot += sum(t.numel() * t.element_size() for t in group)
        return tot
