"""Why did bench.py's instrumented step report 6x the real time for some launches (BENCH_r03.json)?  (dev tool)
Replays bench.py's sequence - timed steps, a host sync, ONE instrumented step - in several modes and prints, per launch kind,
the HIP-event time and the host time spent inside the pair."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd  # noqa: F401
from savit_amd import lib as _lib
from savit_amd.config import get_config
from savit_amd.engine import ViTEngine

cfg = get_config("vit_b_patch16")
B = 128
eng = ViTEngine(cfg, B)
eng.init_params(42)
eng.layout.view(eng.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes) * cfg.embed_dim ** -0.5)
img = torch.randn(B, 224, 224, 3, device="cuda").to(torch.bfloat16)
lab = torch.randint(0, 1000, (B,), device="cuda", dtype=torch.int32)


def train(n):
    for _ in range(n):
        eng.forward(img); eng.loss_backward(lab); eng.optimizer_step(1e-4, 1e-4, 1.0)


def instrumented(precreate, sync_first):
    fwd, bwd = eng._fwd_plan, eng._serial_bwd_plan()
    s = eng._stream()
    n = len(fwd.calls) + len(bwd.calls)
    pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)] if precreate else None
    if precreate:  # make sure the hipEvents exist
        for a, b in pool:
            a.record(); b.record()
        torch.cuda.synchronize()
    if sync_first:
        torch.cuda.synchronize()
    evs, host = [], []
    k = 0

    def run(plan):
        nonlocal k
        for fn, args, label in plan.calls:
            if precreate:
                a, b = pool[k]
            else:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            k += 1
            t0 = time.perf_counter()
            a.record()
            t1 = time.perf_counter()
            rc = fn(*args, s)
            t2 = time.perf_counter()
            b.record()
            t3 = time.perf_counter()
            assert rc == 0
            evs.append((label, a, b))
            host.append((t1 - t0, t2 - t1, t3 - t2))

    w0 = time.perf_counter()
    run(fwd)
    eng.grads.zero_(); eng.loss.zero_()
    _lib.check(eng.L.savit_softmax_xent(eng.logits.data_ptr(), cfg.num_classes, eng.labels.data_ptr(), None, None, 0.1, 1.0 / B,
                                        eng.loss_rows.data_ptr(), eng.loss.data_ptr(), eng.dlogits.data_ptr(), eng.Cp,
                                        eng._off_ptr(eng.grads, "bh"), eng.top1.data_ptr(), eng.top5.data_ptr(), B, cfg.num_classes, s), "xent")
    eng.dres.zero_(); eng.dres_b.zero_()
    run(bwd)
    w1 = time.perf_counter()
    torch.cuda.synchronize()
    w2 = time.perf_counter()
    acc = collections.OrderedDict()
    for (label, a, b), h in zip(evs, host):
        kk = ".".join(label.split(".")[1:]) if label.startswith("l") and label[1].isdigit() else (label[:11] if label.startswith("wgrad.group") else label)
        acc.setdefault(kk, []).append((a.elapsed_time(b) * 1e3,) + tuple(x * 1e6 for x in h))
    return acc, (w1 - w0) * 1e3, (w2 - w0) * 1e3


train(5)
torch.cuda.synchronize()
modes = [("A first call, new events, after sync", False, True), ("B second call, new events, after sync", False, True),
         ("C new events, queue pre-filled (no sync)", False, False), ("D pre-created events, after sync", True, True),
         ("E pre-created events, queue pre-filled", True, False)]
res = {}
for name, pre, sync in modes:
    if not sync:
        train(2)
    acc, issue_ms, wall_ms = instrumented(pre, sync)
    res[name] = acc
    tot = sum(v[0] for vs in acc.values() for v in vs) / 1e3
    print(f"== {name}: sum of event times {tot:.2f} ms, host issue {issue_ms:.2f} ms, wall {wall_ms:.2f} ms")
keys = list(next(iter(res.values())).keys())
print(f"{'launch':18s}" + "".join(f"  {m[0][0]}:ev_us/rec_a/fn/rec_b " for m in modes))
for k in keys:
    row = f"{k:18s}"
    for name, _, _ in modes:
        v = res[name][k]
        row += "  " + "/".join(f"{sum(x[i] for x in v) / len(v):7.1f}" for i in range(4))
    print(row)
# worst single launches of mode A
a = res[modes[0][0]]
flat = [(x[0], k) for k, v in a.items() for x in v]
print("mode A, ten longest single launches:", sorted(flat, reverse=True)[:10])
