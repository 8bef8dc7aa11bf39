"""Ping-pong GEMM (tile 20) vs the pair kernel (tile 13): bitwise equality on every epilogue (same K order), ragged shapes; then rates."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import ops, lib as _lib
bf16 = torch.bfloat16
torch.manual_seed(0)
NEW = int(os.environ.get("PP_TILE", "20"))
def run(M, N, K, epi, tile):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K + epi)
    A = torch.randn(M, K, device="cuda", generator=g).to(bf16); Bt = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(bf16)
    bias = torch.randn(N, device="cuda", generator=g)
    kw = {}
    if epi in (0,): C = torch.empty(M, N, device="cuda", dtype=bf16); kw = dict(bias=bias)
    elif epi == 1: C = torch.empty(M, N, device="cuda", dtype=bf16); kw = dict(bias=bias, C2=torch.empty(M, N, device="cuda", dtype=bf16))
    elif epi == 2: C = torch.empty(M, N, device="cuda"); kw = dict(bias=bias, aux=torch.randn(M, N, device="cuda", generator=g))
    elif epi == 3:
        C = torch.empty(M, N, device="cuda", dtype=bf16)
        kw = dict(aux=torch.randn(M, N, device="cuda", generator=g).to(bf16), colsum=torch.zeros(N, device="cuda"))
    elif epi == 4: C = torch.empty(M, N, device="cuda"); kw = dict(bias=bias)
    ops.gemm_tn(A, Bt, C, epi, tile=tile, **kw)
    torch.cuda.synchronize()
    outs = [C] + [kw[k] for k in ("C2", "colsum") if k in kw]
    return outs
ok = True
for (M, N, K) in [(256, 256, 64), (256, 256, 128), (512, 512, 192), (1000, 768, 768), (25216, 768, 768), (3001, 1000, 256), (300, 2304, 768)]:
    for epi in (0, 1, 2, 3, 4):
        if epi == 3 and N % 8: continue
        a = run(M, N, K, epi, 13); b = run(M, N, K, epi, NEW)
        for i, (x, y) in enumerate(zip(a, b)):
            if epi == 3 and i == 1:  # colsum: atomics, compare loosely
                e = float((x - y).abs().max() / (x.abs().max() + 1e-9)); same = e < 1e-5
            else:
                same = torch.equal(x, y)
            if not same:
                ok = False
                d = (x.float() - y.float()).abs()
                print(f"MISMATCH M{M} N{N} K{K} epi{epi} out{i}: max {float(d.max()):.3e} count {int((d > 0).sum())}")
print("bitwise check:", "OK" if ok else "FAILED", flush=True)
# tail-split pair kernel (tile 18) vs the plain 192x128 kernel (tile 17): bitwise, epilogues without column sums
ok18 = True
for (M, N, K) in [(25216, 768, 768), (25216, 768, 3072), (50432, 384, 1536), (3001, 1000, 256), (100000, 128, 64), (25216, 2304, 768)]:
    for epi in (0, 1, 2, 4):
        a = run(M, N, K, epi, 17); b = run(M, N, K, epi, 18)
        for i, (x, y) in enumerate(zip(a, b)):
            if not torch.equal(x, y):
                ok18 = False
                d = (x.float() - y.float()).abs()
                print(f"MISMATCH18 M{M} N{N} K{K} epi{epi} out{i}: max {float(d.max()):.3e} count {int((d > 0).sum())}")
print("tail-split bitwise check:", "OK" if ok18 else "FAILED", flush=True)
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
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (25216, 2304, 768), (25216, 3072, 768), (25216, 768, 3072), (25216, 768, 768), (25216, 768, 2304), (147712, 1024, 1024), (147712, 4096, 1024)]:
    for tile in (13, 17, 18, NEW):
        us, tf = bench(M, N, K, tile)
        print(f"M{M} N{N} K{K} tile{tile}: {us:8.1f} us {tf:7.1f} TF/s", flush=True)
