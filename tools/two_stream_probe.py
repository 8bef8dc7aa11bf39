"""Timing probe: backward with the weight-gradient GEMMs on a second stream (hazards ignored: timing only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd.config import get_config
from savit_amd.engine import ViTEngine
from savit_amd import lib as _lib

cfg = get_config("vit_b_patch16"); B = 128
eng = ViTEngine(cfg, B); eng.init_params(42)
eng.layout.view(eng.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes) * cfg.embed_dim ** -0.5)
img = torch.randn(B, 224, 224, 3, device="cuda").to(torch.bfloat16)
lab = torch.randint(0, 1000, (B,), device="cuda", dtype=torch.int32)
for _ in range(3):
    eng.forward(img); eng.loss_backward(lab); eng.optimizer_step(1e-4, 1e-4, 1.0)
plan = eng._bwd_plan
side = torch.cuda.Stream()
def run(two):
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    t0 = time.perf_counter()
    for it in range(5):
        for fn, args, label in plan.calls:
            if two and label.endswith(".wgrad"):
                ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
                rc = fn(*args, side.cuda_stream)
            else:
                rc = fn(*args, main.cuda_stream)
            assert rc == 0, label
        ev2 = torch.cuda.Event(); ev2.record(side); main.wait_event(ev2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3
for _ in range(2):
    print("one stream  bwd ms:", round(run(False), 3))
    print("two streams bwd ms:", round(run(True), 3))
