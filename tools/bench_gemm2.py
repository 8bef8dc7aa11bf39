"""GEMM shapes x epilogues x tiles as they occur in one DeiT train step (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import ops

bf16 = torch.bfloat16

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
tiles = [int(t) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 4, 5, 7]
M, F = B * 197, 4 * d
cases = [("qkv fwd", 3 * d, d, 0), ("proj fwd", d, d, 2), ("fc1 fwd", F, d, 1), ("fc2 fwd", d, F, 2),
         ("fc2 dgrad", F, d, 3), ("fc2 dg-nocs", F, d, 3), ("fc1 dgrad", d, F, 0), ("proj dgrad", d, d, 0), ("qkv dgrad", d, 3 * d, 0)]
if os.environ.get("ONLY"):
    cases = [c for c in cases if c[0].startswith(os.environ["ONLY"])]
# flush buffer to defeat infinity-cache residency between iterations
flush = torch.empty(512 * 2 ** 20, dtype=torch.uint8, device="cuda")
for name, N, K, epi in cases:
    A = torch.randn(M, K, device="cuda").to(bf16)
    Bt = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf16)
    kw = {}
    if epi in (0, 1, 3):
        C = torch.empty(M, N, device="cuda", dtype=bf16)
    else:
        C = torch.empty(M, N, device="cuda")
        kw["aux"] = torch.randn(M, N, device="cuda")
    if epi == 1:
        kw["C2"] = torch.empty(M, N, device="cuda", dtype=bf16); kw["bias"] = torch.randn(N, device="cuda")
    if epi == 3:
        kw["aux"] = torch.randn(M, N, device="cuda").to(bf16)
        if "nocs" not in name:
            kw["colsum"] = torch.zeros(N, device="cuda")
    fl = 2.0 * M * N * K
    res = []
    for tile in tiles:
        def run():
            flush.zero_()
            ops.gemm_tn(A, Bt, C, epi, tile=tile, **kw)
        def base():
            flush.zero_()
        t = timeit(run) - timeit(base)
        res.append(f"t{tile}: {t*1e6:6.1f}us {fl/t/1e12:6.0f}TF")
    print(f"{name:11s} N={N:4d} K={K:4d} epi={epi} | " + " | ".join(res))
