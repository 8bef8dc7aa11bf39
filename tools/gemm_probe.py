import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import lib as _l
if os.environ.get("SAVIT_EXP_LIB"):
    _l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "exp", "libsavit_%s.so" % os.environ["SAVIT_EXP_LIB"])
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
for (M, N, K) in [(25216, 2304, 768), (25216, 3072, 768), (25216, 768, 3072), (25216, 768, 768), (65536, 2304, 768)]:
    for tile in (12, 13, 17):
        us, tf = bench(M, N, K, tile)
        print(f"lib={os.environ.get('SAVIT_EXP_LIB','base')} probe={os.environ.get('SAVIT_PROBE_L2','0')} M{M} N{N} K{K} tile{tile}: {us:8.1f} us {tf:7.1f} TF/s", flush=True)
