"""Micro-benchmark of the GEMM kernels on the DeiT-B / DeiT-S shapes (dev tool, not the judged bench)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import ops

bf16 = torch.bfloat16

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    M = B * 197
    shapes = [("qkv", M, 3 * d, d), ("proj", M, d, d), ("fc1", M, 4 * d, d), ("fc2", M, d, 4 * d)]
    for name, M_, N, K in shapes:
        A = torch.randn(M_, K, device="cuda").to(bf16)
        Bt = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf16)
        C = torch.empty(M_, N, device="cuda", dtype=bf16)
        fl = 2.0 * M_ * N * K
        for tile in (7, 12, 13, 17):
            t = timeit(lambda: ops.gemm_tn(A, Bt, C, 0, tile=tile))
            print(f"fwd  {name:5s} M={M_} N={N} K={K} tile={tile}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
        # torch (hipBLASLt) reference for context
        W = Bt.t().contiguous()
        t = timeit(lambda: torch.matmul(A, W))
        print(f"     {name:5s} torch.matmul            : {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
        dW = torch.zeros(K, N, device="cuda")
        for sp in (0,):
            t = timeit(lambda: ops.gemm_wgrad(A, C, dW, splits=sp))
            print(f"wgrad {name:5s} splits={sp}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
        At = A.t().contiguous()
        t = timeit(lambda: torch.matmul(At, C))
        print(f"     {name:5s} torch wgrad matmul(A^T,C): {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
    # layernorm
    x = torch.randn(M, d, device="cuda")
    g = torch.ones(d, device="cuda"); b = torch.zeros(d, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, g, b)
    t = timeit(lambda: ops.layernorm_fwd(x, g, b, out=y, mean=mean, rstd=rstd))
    print(f"ln fwd: {t*1e6:.1f} us  {(6*M*d+8*M)/t/1e9:.0f} GB/s")
    dy = torch.randn(M, d, device="cuda").to(bf16); dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
    dx = torch.empty(M, d, device="cuda"); dxb = torch.empty(M, d, device="cuda", dtype=bf16); dres = torch.randn(M, d, device="cuda")
    t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres_in=dres, dx=dx, dx_bf16=dxb, dcolsum=dg))
    print(f"ln bwd: {t*1e6:.1f} us  {(16*M*d)/t/1e9:.0f} GB/s")

main()
