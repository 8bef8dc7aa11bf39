"""Timing probe: two half-batch engines on two streams vs one full-batch engine (fwd + bwd only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd.config import get_config
from savit_amd.engine import ViTEngine

cfg = get_config("vit_b_patch16")
def mk(B):
    e = ViTEngine(cfg, B); e.init_params(42)
    e.layout.view(e.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes) * cfg.embed_dim ** -0.5)
    img = torch.randn(B, 224, 224, 3, device="cuda").to(torch.bfloat16)
    lab = torch.randint(0, 1000, (B,), device="cuda", dtype=torch.int32)
    return e, img, lab
full = mk(128)
halves = [mk(64), mk(64)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def step_full():
    e, img, lab = full
    e.forward(img); e.loss_backward(lab)
def step_halves():
    for (e, img, lab), s in zip(halves, streams):
        with torch.cuda.stream(s):
            e.forward(img)
    for (e, img, lab), s in zip(halves, streams):
        with torch.cuda.stream(s):
            e.loss_backward(lab)
def step_halves_seq():
    for (e, img, lab) in halves:
        e.forward(img); e.loss_backward(lab)
def timeit(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for ov in (True, False):
    for e in [full[0]] + [h[0] for h in halves]: e.overlap_wgrad = ov
    print("overlap_wgrad", ov, "full B=128: %.3f ms" % timeit(step_full), " 2xB=64 two streams: %.3f ms" % timeit(step_halves),
          " 2xB=64 sequential: %.3f ms" % timeit(step_halves_seq))
