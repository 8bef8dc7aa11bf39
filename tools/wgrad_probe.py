"""Weight-gradient GEMM timing over K-splits for one shape (dev tool): python tools/wgrad_probe.py M Kin Nout"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd  # noqa: F401
from savit_amd import ops
M, Kin, Nout = (int(v) for v in sys.argv[1:4])
X = torch.randn(M, Kin, device="cuda").to(torch.bfloat16)
dY = torch.randn(M, Nout, device="cuda").to(torch.bfloat16)
dW = torch.zeros(Kin, Nout, device="cuda")
for sp in (0, 8, 16, 32, 48, 64, 96, 128, 192, 256):
    for _ in range(3):
        ops.gemm_wgrad(X, dY, dW, splits=sp)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.gemm_wgrad(X, dY, dW, splits=sp)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"M{M} Kin{Kin} Nout{Nout} splits {sp:4d}: {us:7.1f} us  {2.0*M*Kin*Nout/us/1e6:7.1f} TF/s  {(M*(Kin+Nout)*2)/us/1e6:6.2f} TB/s", flush=True)
