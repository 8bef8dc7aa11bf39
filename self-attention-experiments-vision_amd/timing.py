"""Per-launch timing of the engines' launch plans (the measurement side of the hot path).

`LaunchTimer` brackets selected launches with HIP events from a pre-created pool (savit_timer_*: events made with
hipEventDisableSystemFence, so that recording one does not write back and invalidate the caches in front of the launch being
timed - the default-flag events of round 3 did, and the bench line of that round carried one kernel class at 6x its rocprofv3
time on some boxes).  Two uses:

* bench.py's timed region: only the launches of the dominant kernel are bracketed, over ALL timed steps, while the host runs far
  ahead of the GPU - the `roofline` object of the JSON line.
* `instrumented_steps`: every launch of a step is bracketed, the whole step enqueued behind a gate kernel (savit_spin) so that no
  pair can contain host time, several repetitions, per-label minimum - the per-class breakdown, and the check that it adds up.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import torch

from . import lib as _lib


class LaunchTimer:
    """A pool of `pairs` (begin, end) timing events; `only` = the launch labels to bracket (None: every launch)."""

    def __init__(self, pairs: int, only: Optional[Iterable[str]] = None):
        self.L = _lib.load()
        self.n = int(pairs)
        h = ctypes.c_void_p()
        _lib.check(self.L.savit_timer_create(2 * self.n, ctypes.byref(h)), "savit_timer_create")
        self.h = h
        self.only = None if only is None else frozenset(only)
        self.used: List[str] = []
        self.dropped = 0  # launches that wanted a pair when the pool was empty

    def reset(self):
        self.used = []
        self.dropped = 0

    def begin(self, label: str, stream: int) -> int:
        """-> pair index to pass to end(), or -1 when this launch is not bracketed"""
        if self.only is not None and label not in self.only:
            return -1
        i = len(self.used)
        if i >= self.n:
            self.dropped += 1
            return -1
        self.used.append(label)
        self.L.savit_timer_record(self.h, 2 * i, stream)
        return i

    def end(self, i: int, stream: int):
        self.L.savit_timer_record(self.h, 2 * i + 1, stream)

    def results(self) -> List[Tuple[str, float]]:
        """[(label, milliseconds)] in issue order.  The stream(s) must have been synchronised."""
        ms = ctypes.c_float()
        out = []
        for i, label in enumerate(self.used):
            _lib.check(self.L.savit_timer_elapsed_ms(self.h, 2 * i, 2 * i + 1, ctypes.byref(ms)), "savit_timer_elapsed_ms")
            out.append((label, float(ms.value)))
        return out

    def span_ms(self) -> float:
        """first begin -> last end of what was recorded (same stream)"""
        ms = ctypes.c_float()
        _lib.check(self.L.savit_timer_elapsed_ms(self.h, 0, 2 * (len(self.used) - 1) + 1, ctypes.byref(ms)), "savit_timer_elapsed_ms")
        return float(ms.value)

    def close(self):
        if self.h is not None:
            self.L.savit_timer_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def timed_call(timer: Optional[LaunchTimer], label: str, fn: Callable, *args) -> int:
    """fn(*args) with args[-1] the stream, bracketed when `timer` tracks `label`; raises on a non-zero return."""
    if timer is None:
        rc = fn(*args)
    else:
        k = timer.begin(label, args[-1])
        rc = fn(*args)
        if k >= 0:
            timer.end(k, args[-1])
    if rc != 0:
        _lib.check(rc, label)
    return rc


def instrumented_steps(eng, run_step: Callable[[], None], reps: int = 3, gate_us: int = 0) -> dict:
    """Run `run_step()` (the engine's ordinary forward / loss_backward / optimizer_step calls) `reps` times with EVERY launch
    bracketed, each repetition enqueued behind a gate kernel.  One stream: engines that normally put weight gradients on side
    streams run their serial plan here (per-kernel figures are taken serially: DESIGN.md 6).

    -> {"labels": {label: min ms over the repetitions}, "reps": [{"sum_ms", "span_ms", "host_issue_ms"}], "gate_us", "launches",
        "pair_overhead_ms": median duration of an EMPTY bracket (what every figure above contains on top of its kernel)}"""
    import time

    L = _lib.load()
    saved_timer, saved_overlap = eng.launch_timer, eng.overlap_wgrad
    eng.overlap_wgrad = False
    s = torch.cuda.current_stream().cuda_stream
    timer = None
    try:
        # a first, untimed pass counts the launches (and builds whatever plan is not built yet)
        probe = _CountingTimer()
        eng.launch_timer = probe
        run_step()
        torch.cuda.synchronize()
        n = probe.count
        timer = LaunchTimer(n + 8)
        eng.launch_timer = timer
        # warm the event pool (first record of an event may allocate) and learn the host time of one instrumented issue
        timer.reset()
        t0 = time.perf_counter()
        run_step()
        issue_ms = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        if gate_us <= 0:
            gate_us = int(min(150000, max(3000, 2.0 * issue_ms * 1e3)))  # (1.5x / 2 ms until round 5: one slow host iteration on a shared box failed the gate)
        best: Dict[str, float] = {}
        order: List[str] = []
        rep_info = []
        retries = 0
        for _ in range(reps):
            # A repetition whose host issue outlasted the gate (another tenant's burst on a shared box) is not a measurement of the
            # GPU: it is run again behind a gate twice as long, at most three times - what is reported comes from gated repetitions only
            # (a repetition that still misses is reported with gate_reached false, and bench.py marks the line invalid).
            for attempt in range(4):
                timer.reset()
                torch.cuda.synchronize()
                _lib.check(L.savit_spin(gate_us, s), "savit_spin")
                t0 = time.perf_counter()
                run_step()
                host = (time.perf_counter() - t0) * 1e3
                torch.cuda.synchronize()
                res = timer.results()
                if timer.dropped:
                    raise RuntimeError("instrumented step issued more launches than its counting pass")
                if host * 1e3 < gate_us or attempt == 3:
                    break
                retries += 1
                gate_us = int(min(150000, 2 * gate_us))
            seen: Dict[str, int] = {}
            tot = 0.0
            for label, ms in res:
                k = seen.get(label, 0)
                seen[label] = k + 1
                key = label if k == 0 else f"{label}#{k}"  # a label launched twice in a step (refresh casts) stays two rows
                if key not in best:
                    order.append(key)
                    best[key] = ms
                else:
                    best[key] = min(best[key], ms)
                tot += ms
            rep_info.append({"sum_ms": tot, "span_ms": timer.span_ms(), "host_issue_ms": host, "gate_reached": host * 1e3 < gate_us})
        # what a bracket costs by itself: pairs with nothing between them (a marker is a packet of its own on the queue; under
        # rocprofv3 every dispatch carries more)
        timer.reset()
        torch.cuda.synchronize()
        for _ in range(min(32, timer.n)):
            k = timer.begin("empty", s)
            timer.end(k, s)
        torch.cuda.synchronize()
        empties = sorted(ms for _, ms in timer.results())
        return {"labels": {k: best[k] for k in order}, "reps": rep_info, "gate_us": gate_us, "launches": n, "gate_retries": retries,
                "pair_overhead_ms": empties[len(empties) // 2]}
    finally:
        eng.launch_timer, eng.overlap_wgrad = saved_timer, saved_overlap
        if timer is not None:
            timer.close()


class _CountingTimer:
    """Stands in for a LaunchTimer to count the launches of a step."""

    only = None

    def __init__(self):
        self.count = 0

    def begin(self, label, stream):
        self.count += 1
        return -1

    def end(self, i, stream):
        pass
