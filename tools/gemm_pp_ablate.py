import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import ops
bf16 = torch.bfloat16
def bench(M, N, K, tile, n=20):
    A = torch.randn(M, K, device="cuda").to(bf16); Bt = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf16)
    C = torch.empty(M, N, device="cuda", dtype=bf16)
    for _ in range(3): ops.gemm_tn(A, Bt, C, 0, tile=tile)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): ops.gemm_tn(A, Bt, C, 0, tile=tile)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    return ms * 1e3, 2.0 * M * N * K / ms / 1e9
names = {20: "full", 101: "no DMA", 102: "no LDS reads", 103: "no DMA, no reads", 104: "no barriers", 107: "MFMA only", 108: "no MFMA", 109: "no MFMA no DMA (reads+barriers)", 110: "no MFMA no reads (DMA+barriers)"}
for (M, N, K) in [(4096, 4096, 4096), (25216, 3072, 768), (25216, 768, 3072)]:
    for rep in range(2):
        for tile in (20, 101, 102, 103, 104, 107, 108, 109, 110):
            us, tf = bench(M, N, K, tile)
            print(f"M{M} N{N} K{K} {names[tile]:32s}: {us:8.1f} us {tf:7.1f} TF/s", flush=True)
