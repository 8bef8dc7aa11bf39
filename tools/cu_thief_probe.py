"""What does a resident all-reduce cost the train step, and does planning for it help?  (row a16 / e; no 8-GPU node is available)
A "CU thief" - savit_hold_cus: n one-wave workgroups that each keep 96 KB of a CU's LDS, so no large-tile GEMM workgroup is placed
there - runs on a second stream for the length of every step, standing in for the CUs RCCL's channels hold during backward.  The step
is timed with the engine planned for the whole chip (reserved_cus = 0) and planned for the remaining CUs (reserved_cus = n): grouped
weight-gradient launches of one tile per REMAINING CU and the TN tile choice priced for that many CUs.
usage: python tools/cu_thief_probe.py [model batch]   -> a table (also JSON on the last line)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd  # noqa: F401
from savit_amd import lib as _lib
from savit_amd.config import get_config
from savit_amd.engine import ViTEngine

model = sys.argv[1] if len(sys.argv) > 1 else "vit_b_patch16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cfg = get_config(model)
L = _lib.load()
img = torch.randn(B, 224, 224, 3, device="cuda").to(torch.bfloat16)
lab = torch.randint(0, 1000, (B,), device="cuda", dtype=torch.int32)
side = torch.cuda.Stream()


def run(reserved, thief, steps=20, warmup=5, hold_us=0):
    eng = ViTEngine(cfg, B, reserved_cus=reserved)
    eng.init_params(42)
    eng.layout.view(eng.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes) * cfg.embed_dim ** -0.5)
    ev = torch.cuda.Event()
    main = torch.cuda.current_stream()

    def step():
        if thief:
            ev.record(main)
            side.wait_event(ev)
            _lib.check(L.savit_hold_cus(thief, hold_us, side.cuda_stream), "savit_hold_cus")
        eng.forward(img)
        eng.loss_backward(lab)
        eng.optimizer_step(1e-4, 1e-4, 1.0)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        if thief:
            main.wait_stream(side)  # the next step's thief starts with the next step, not behind this one
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    groups = sorted(k for k in getattr(eng, "group_flops", {}))
    del eng
    torch.cuda.empty_cache()
    return ms, len(groups)


run(0, 0, steps=10)  # the first engine of a process has been seen 10 % slower than every later one (three boxes): not the baseline
base = min(run(0, 0)[0], run(0, 0)[0])
rows = [{"thief_cus": 0, "reserved_cus": 0, "ms_per_step": round(base, 3), "vs_alone": 1.0}]
print(f"{model} B={B}: alone {base:.3f} ms/step")
for n in (16, 32):
    hold = int(base * 1e3 * 0.97)  # the thief holds its CUs for (almost) the whole step
    for r in (0, n):
        ms, ng = run(r, n, hold_us=hold)
        rows.append({"thief_cus": n, "reserved_cus": r, "ms_per_step": round(ms, 3), "vs_alone": round(ms / base, 4), "grouped_wgrad_launches": ng})
        print(f"thief {n:3d} CUs, engine planned for {256 - r:3d} CUs (reserved_cus {r:2d}): {ms:.3f} ms/step = {ms / base:.3f} x alone   ({ng} grouped weight-gradient launches)", flush=True)
    ms, ng = run(n, 0)
    rows.append({"thief_cus": 0, "reserved_cus": n, "ms_per_step": round(ms, 3), "vs_alone": round(ms / base, 4)})
    print(f"no thief, reserved_cus {n:2d}: {ms:.3f} ms/step = {ms / base:.3f} x alone (what planning for an all-reduce costs when none is resident)", flush=True)
print(json.dumps({"model": model, "batch": B, "rows": rows}))
