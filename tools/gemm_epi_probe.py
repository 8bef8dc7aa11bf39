import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import lib as _l
if os.environ.get("SAVIT_EXP_LIB"):
    _l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "exp", "libsavit_%s.so" % os.environ["SAVIT_EXP_LIB"])
from savit_amd import ops
bf16 = torch.bfloat16
def bench(M, N, K, tile, epi, n=20, cold=True):
    A = torch.randn(M, K, device="cuda").to(bf16); Bt = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf16)
    kw = {}
    if epi == 0: C = torch.empty(M, N, device="cuda", dtype=bf16)
    elif epi == 2: C = torch.empty(M, N, device="cuda"); kw["aux"] = torch.randn(M, N, device="cuda")
    elif epi == 1: C = torch.empty(M, N, device="cuda", dtype=bf16); kw["C2"] = torch.empty_like(C); kw["bias"] = torch.randn(N, device="cuda")
    elif epi == 3: C = torch.empty(M, N, device="cuda", dtype=bf16); kw["aux"] = torch.randn(M, N, device="cuda").to(bf16); kw["colsum"] = torch.zeros(N, device="cuda")
    trash = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device="cuda")
    ts = []
    for _ in range(n):
        if cold: trash.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.gemm_tn(A, Bt, C, epi, tile=tile, **kw); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort(); ms = ts[len(ts) // 2]
    return ms * 1e3, 2.0 * M * N * K / ms / 1e9
shapes = [(25216, 768, 3072), (25216, 768, 768), (25216, 3072, 768)]
for (M, N, K) in shapes:
    for epi in (0, 1, 2, 3):
        for tile in (12, 13):
            us, tf = bench(M, N, K, tile, epi)
            print(f"lib={os.environ.get('SAVIT_EXP_LIB','base')} M{M} N{N} K{K} epi{epi} tile{tile}: {us:8.1f} us {tf:7.1f} TF/s", flush=True)
