"""Fused-epilogue GEMM timings on the DeiT-B shapes per tile variant (dev tool).  Usage: gemm_epi_bench.py tile [tile ...]
(SAVIT_EPI_SHAPE="M,d,F" selects another model: e.g. 50432,384,1536 = DeiT-S at 256 images; tile 0 = the library's own choice)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import ops
bf16 = torch.bfloat16
M, d, F = (int(v) for v in os.environ.get("SAVIT_EPI_SHAPE", "25216,768,3072").split(","))
def mk(epi, N, K):
    A = torch.randn(M, K, device="cuda").to(bf16); Bt = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf16)
    bias = torch.randn(N, device="cuda")
    if epi == 0: return A, Bt, torch.empty(M, N, device="cuda", dtype=bf16), {}
    if epi == 1: return A, Bt, torch.empty(M, N, device="cuda", dtype=bf16), dict(bias=bias, C2=torch.empty(M, N, device="cuda", dtype=bf16))
    if epi == 2: return A, Bt, torch.empty(M, N, device="cuda"), dict(bias=bias, aux=torch.randn(M, N, device="cuda"))
    if epi == 3: return A, Bt, torch.empty(M, N, device="cuda", dtype=bf16), dict(aux=torch.randn(M, N, device="cuda").to(bf16), colsum=torch.zeros(N, device="cuda"))
cases = [("qkv", 0, 3 * d, d), ("proj+res", 2, d, d), ("fc1+gelu", 1, F, d), ("fc2+res", 2, d, F), ("fc2.dgrad+dgelu", 3, F, d), ("fc1.dgrad", 0, d, F), ("qkv.dgrad", 0, d, 3 * d)]
tiles = [int(t) for t in sys.argv[1:]] or [13, 17, 20]
for name, epi, N, K in cases:
    A, Bt, C, kw = mk(epi, N, K)
    junk = torch.empty(64 * 2 ** 20, device="cuda")  # 256 MB: flush the memory-side cache between launches like the pipeline does
    for tile in tiles:
        ts = []
        if epi == 3:  # column sums as the per-row-tile slab of this tile (what the engine passes)
            kw["colsum"] = torch.zeros(max(1, ops.gemm_colsum_rows(M, N, K, tile)), N, device="cuda")
        try:
            for rep in range(12):
                junk.fill_(1.0)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); ops.gemm_tn(A, Bt, C, epi, tile=tile, **kw); b.record(); torch.cuda.synchronize()
                if rep >= 2: ts.append(a.elapsed_time(b) * 1e3)
        except ValueError:
            print(f"{name:16s} N{N} K{K} tile{tile}: not served by this tile", flush=True)
            continue
        ts.sort()
        print(f"{name:16s} N{N} K{K} tile{tile}: median {ts[len(ts)//2]:7.1f} us  min {ts[0]:7.1f} us   {2.0*M*N*K/ts[len(ts)//2]/1e6:7.1f} TF/s  [{os.environ.get('SAVIT_PP_PHASES','-')}/{os.environ.get('SAVIT_PP_SLEEP','-')}]", flush=True)
