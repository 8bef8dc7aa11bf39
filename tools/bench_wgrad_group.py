"""Grouped weight-gradient launch (savit_gemm_bf16_wgrad_grouped: one workgroup per 256x256 output tile over all tokens) against the
per-weight launches it replaces (token range split over workgroups, partial slabs + ordered reduce) on the shapes of `layers`
encoder layers.  python tools/bench_wgrad_group.py [d F M layers]      (default: DeiT-B/16 at 128 images, 2 layers)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import savit_amd  # noqa: F401
from savit_amd import lib as _l, ops

if os.environ.get("SAVIT_EXP_LIB"):
    _l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "exp", "libsavit_%s.so" % os.environ["SAVIT_EXP_LIB"])

bf16 = torch.bfloat16
d, F, M, layers = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (768, 3072, 25216, 2)))
tile = int(os.environ.get("SAVIT_GROUP_TILE", "0")) or (256 if d % 256 == 0 and F % 256 == 0 else (384 if d % 384 == 0 and F % 384 == 0 else 128))
g = torch.Generator(device="cuda").manual_seed(0)
probs = []
for _ in range(layers):
    for Kin, Nout in ((F, d), (d, F), (d, d), (d, 3 * d)):
        X = torch.randn(M, Kin, device="cuda", generator=g).to(bf16)
        dY = torch.randn(M, Nout, device="cuda", generator=g).to(bf16)
        probs.append((X, dY, torch.zeros(Kin, Nout, device="cuda")))
flops = sum(2.0 * M * p[2].shape[0] * p[2].shape[1] for p in probs)
ws = {(p[2].shape): ops.wgrad_workspace(M, p[2].shape[0], p[2].shape[1]) for p in probs}


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def single():
    for X, dY, dW in probs:
        ops.gemm_wgrad(X, dY, dW, workspace=ws[dW.shape])


def grouped():
    for i in range(0, len(probs), 16):
        ops.gemm_wgrad_grouped(probs[i:i + 16], tile=tile)


for rnd in range(3):  # interleaved rounds in one process
    ts, tg = timeit(single), timeit(grouped)
    print(f"round {rnd}: per-weight launches {ts:8.1f} us = {flops / ts / 1e6:7.1f} TF/s | grouped (tile {tile}) {tg:8.1f} us = {flops / tg / 1e6:7.1f} TF/s "
          f"| d={d} F={F} M={M} layers={layers}", flush=True)
# same sums?
for X, dY, dW in probs:
    dW.zero_()
single()
a = [p[2].clone() for p in probs]
for X, dY, dW in probs:
    dW.zero_()
grouped()
err = max(float((x - p[2]).norm() / x.norm()) for x, p in zip(a, probs))
print(f"grouped vs per-weight results: max rel-L2 difference {err:.2e} (fp32 summation order)")
