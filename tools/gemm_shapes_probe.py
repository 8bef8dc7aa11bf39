"""GEMM TN kernel rates on square and DeiT-B shapes per tile variant (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import ops
bf16 = torch.bfloat16
def bench(M, N, K, tile, epi=0, n=20):
    A = torch.randn(M, K, device="cuda").to(bf16); Bt = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf16)
    C = torch.empty(M, N, device="cuda", dtype=bf16)
    for _ in range(3): ops.gemm_tn(A, Bt, C, epi, tile=tile)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): ops.gemm_tn(A, Bt, C, epi, tile=tile)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    return ms * 1e3, 2.0 * M * N * K / ms / 1e9
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (25216, 2304, 768), (25216, 3072, 768), (25216, 768, 3072), (25216, 768, 768), (25216, 768, 2304)]
for (M, N, K) in shapes:
    for tile in (12, 13, 17):
        us, tf = bench(M, N, K, tile)
        print(f"M{M} N{N} K{K} tile{tile}: {us:8.1f} us {tf:7.1f} TF/s", flush=True)
    W = (torch.randn(K, N, device="cuda") / K ** 0.5).to(bf16); A = torch.randn(M, K, device="cuda").to(bf16)
    for _ in range(3): torch.matmul(A, W)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): torch.matmul(A, W)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"M{M} N{N} K{K} hipBLASLt: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
