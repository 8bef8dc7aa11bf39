"""Data-parallel gradient exchange: the MI355X counterpart of `jax.lax.pmean(grads, 'batch')`
(/root/reference/train.py:96; replicate/pmap at :228-230).

One process per GPU.  Gradients live in ONE flat fp32 buffer laid out layer-major (engine.ParamLayout), and
backward finishes layers in reverse order, so a bucket is a contiguous slice [start, end) that becomes final
right after a known launch of the backward plan.  Each bucket is all-reduced (SUM) with torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests) as soon as it is final, asynchronously, so
the exchange overlaps the remaining backward GEMMs; the 1/world factor is folded into the fused AdamW kernel
(grad_scale) instead of a separate pass.  Buckets are sized for xGMI's point-to-point links:
default >= 48 MB so each ring step moves multi-MB chunks per link."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


PIN_ENV = "SAVIT_PIN_RCCL_CHANNELS"


def default_reserved_cus(world: int, env=None, n_cus: int = 256) -> int:
    """CUs a rank plans to leave to the resident RCCL all-reduce during backward (engine `reserved_cus`): 0 alone; at world > 1
    SAVIT_RESERVED_CUS, else the channel bound the user gave RCCL (NCCL_MAX_NCHANNELS: one channel = one workgroup = one CU; clamped to
    n_cus - 1 so that a generous bound like 512 cannot make the engine's range check raise), else 16.
    Why plan at all (tools/cu_thief_probe.py, profiles/r04_cu_thief.log: a stand-in that holds n CUs for the length of the step,
    DeiT-B/16 at 128 images): with 16 CUs taken the step is 18 % slower when the launch plan assumes the whole chip (a grid of one tile
    per CU runs a second, nearly empty round) and 6 % slower when it was planned for 240 CUs; with 32 taken 27 % against 17 %.
    UNVERIFIED ON MORE THAN ONE GPU: the 16 is a planning constant measured against a stand-in, never against RCCL over xGMI (no 8-GPU
    node was available in any round); `bench.py --gpus N` therefore also times reserved_cus 0 / 8 / 16 / 32 in the same job and prints
    all four (`reserved_cus_sweep`)."""
    import os

    env = os.environ if env is None else env
    if env.get("SAVIT_RESERVED_CUS"):
        return int(env["SAVIT_RESERVED_CUS"])
    if world > 1 and env.get("NCCL_MAX_NCHANNELS"):
        return max(0, min(int(env["NCCL_MAX_NCHANNELS"]), n_cus - 1))  # the user bounded RCCL: plan for exactly that
    return 16 if world > 1 else 0


def rccl_channel_env(reserved_cus: int, env=None, pin: Optional[bool] = None) -> Dict[str, str]:
    """OPT-IN (SAVIT_PIN_RCCL_CHANNELS=1, or pin=True): the RCCL settings that keep its resident kernels inside `reserved_cus` CUs - an
    RCCL channel is one workgroup that owns a CU for the length of the collective, so NCCL_MAX_NCHANNELS = NCCL_MIN_NCHANNELS =
    reserved_cus pins the footprint to the plan.  Round 5 applied this to every multi-GPU run; no N > 1 measurement backs it (RCCL's own
    choice on an 8-GPU xGMI node is 28-64 channels, and forcing 16 may cost all-reduce bandwidth and RAISE `allreduce_exposed_ms` - the
    very thing the plan is meant to lower), so the default is RCCL's own choice again until one N > 1 bench compares pinned against
    unpinned.  Values the user already exported win (and `default_reserved_cus` then plans for THEIR bound).  Must be in the
    environment before the first RCCL communicator is created: bench.launch_ranks / train.py export it to the rank processes,
    `apply_rccl_channel_env` sets it in a rank started by an external launcher."""
    import os

    env = os.environ if env is None else env
    if pin is None:
        pin = env.get(PIN_ENV, "0") == "1"
    out: Dict[str, str] = {}
    if not pin or reserved_cus <= 0:
        return out
    mx = env.get("NCCL_MAX_NCHANNELS") or str(int(reserved_cus))
    out["NCCL_MAX_NCHANNELS"] = mx
    out["NCCL_MIN_NCHANNELS"] = env.get("NCCL_MIN_NCHANNELS") or str(min(int(mx), int(reserved_cus)))
    return out


def apply_rccl_channel_env(reserved_cus: int, pin: Optional[bool] = None) -> Dict[str, str]:
    """Export `rccl_channel_env` into this process (call BEFORE dist.init_process_group).  -> what was set ({} unless pinning is on)."""
    import os

    e = rccl_channel_env(reserved_cus, pin=pin)
    os.environ.update(e)
    return e


def rccl_channels_in_effect(env=None) -> Optional[int]:
    """The channel bound RCCL runs under in this process (None: RCCL's own choice) - recorded in the N > 1 bench line."""
    import os

    env = os.environ if env is None else env
    v = env.get("NCCL_MAX_NCHANNELS")
    return int(v) if v else None


def parse_rccl_channels(debug_text: str) -> Optional[int]:
    """Number of collective channels RCCL reports under NCCL_DEBUG=INFO (max over communicators), or None when the log has no such
    line.  RCCL prints e.g. 'NCCL INFO 16 coll channels, 0 collnet channels, 0 nvls channels, 16 p2p channels, ...' per communicator
    and 'Channel 07/16 : 0' lines while it builds the rings."""
    import re

    n = [int(m.group(1)) for m in re.finditer(r"(\d+) coll channels", debug_text)]
    n += [int(m.group(1)) for m in re.finditer(r"Channel \d+/(\d+)\s*:", debug_text)]
    return max(n) if n else None


def plan_buckets(layer_starts: List[int], final_start: int, total: int, min_bucket_elems: int) -> List[Tuple[int, int, str]]:
    """Buckets in the order backward completes them: [(start, end, trigger_label)].  trigger_label names the
    backward launch after which the slice is final: 'l{i}.ln1.bwd' closes layer i, 'Wpe.wgrad' closes the
    embedding block (and everything still open)."""
    L = len(layer_starts)
    buckets: List[Tuple[int, int, str]] = []
    end = total
    i = L - 1
    # The LAST bucket's exchange has no backward work left to hide behind, so it is kept as small as the layout allows:
    # the embedding block plus layer 0 only.  Whatever is still open above layer 0 closes as its own (possibly short)
    # bucket at 'l1.ln1.bwd' and overlaps layer 0's backward.
    while i >= 1:
        j = i
        while j > 1 and end - layer_starts[j] < min_bucket_elems:
            j -= 1
        buckets.append((layer_starts[j], end, f"l{j}.ln1.bwd"))
        end = layer_starts[j]
        i = j - 1
    buckets.append((0, end, "Wpe.wgrad"))
    assert buckets[-1][0] == 0 and sum(e - s for s, e, _ in buckets) == total
    return buckets


def plan_buckets_for(layout, min_bucket_elems: int) -> List[Tuple[int, int, str]]:
    """Buckets for an engine layout (engine.ParamLayout or cait_engine.CaiTLayout).  CaiT: the token-only layers and the final
    block sit at the end of the buffer and finish first (after 'cls.grad'); the SA layers follow as for ViT."""
    ca = getattr(layout, "ca_start", None)
    if ca:
        tail = (ca[0], layout.total, "cls.grad")
        rest = plan_buckets(layout.layer_start, ca[0], ca[0], min_bucket_elems)
        return [tail] + rest
    return plan_buckets(layout.layer_start, layout.final_start, layout.total, min_bucket_elems)


class GradSync:
    def __init__(self, flat_grads: torch.Tensor, buckets: List[Tuple[int, int, str]], group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.flat = flat_grads
        self.buckets = buckets
        self.group = group
        self.world = dist.get_world_size(group)
        self.works: List = []
        self.launched: List[int] = []

    def hooks(self) -> Dict[str, Callable[[], None]]:
        return {label: (lambda i=i: self._launch(i)) for i, (_, _, label) in enumerate(self.buckets)}

    def _launch(self, i: int):
        s, e, _ = self.buckets[i]
        self.launched.append(i)
        self.works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        """Block the current stream (GPU) / thread (CPU) until every launched bucket is reduced."""
        if sorted(self.launched) != list(range(len(self.buckets))):
            raise RuntimeError(f"gradient buckets launched {self.launched}, expected all {len(self.buckets)}")
        for w in self.works:
            w.wait()
        self.works.clear()
        self.launched.clear()

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def broadcast_params(flat_params: torch.Tensor, src: int = 0, group=None):
    """flax.jax_utils.replicate (train.py:228): every rank starts from rank 0's parameters."""
    dist.broadcast(flat_params, src=src, group=group)


def allreduce_scalar_mean(t: torch.Tensor, group=None) -> torch.Tensor:
    """jax.lax.psum(loss)/n for logging (train.py:119-120)."""
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t /= dist.get_world_size(group)
    return t
