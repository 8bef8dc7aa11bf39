"""Bitwise repeatability of whole forward passes (no atomics on that path) - run two copies at once to exercise GPU time-slicing
(see tools/kernel_det_probe.py).  Usage: python tools/forward_det_probe.py [iterations]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd  # noqa: F401
from savit_amd.model import create_model
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, B in (("vit_s_patch16", 16), ("cait_xxs_24", 16), ("mixer_s_patch16", 16), ("tnt_b_patch16", 8)):
    m = create_model(name, dtype=torch.bfloat16)
    e = m.engine(B)
    e.init_params(3)
    e.layout.view(e.params, "Wh").normal_(0, 0.02)
    e.weights_stale = True
    x = torch.randn(B, 224, 224, 3, device="cuda").to(torch.bfloat16)
    ref = e.forward(x).clone()
    bad = 0
    for it in range(n_it):
        out = e.forward(x)
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            print(name, "iteration", it, "differs: max abs", float((out - ref).abs().max()), flush=True)
    print(name, "mismatches", bad, "of", n_it, flush=True)
    del e, m
    torch.cuda.empty_cache()
