"""How warm is a GEMM operand that the previous kernel wrote?  (MALL / L2 residency experiment)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import ops
bf16 = torch.bfloat16
M, N, K = 25216, 768, 3072
A = torch.randn(M, K, device="cuda").to(bf16); A2 = A.clone(); U = torch.empty_like(A); U2 = torch.randn(M, K, device="cuda").to(bf16)
Bt = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf16)
C = torch.empty(M, N, device="cuda", dtype=bf16)
trash = torch.empty(768 * 1024 * 1024, dtype=torch.uint8, device="cuda")
TILE = int(os.environ.get("TILE", "12"))
def run(prep, tile=None, n=10, rev=False):
    tile = tile or TILE
    ts = []
    for _ in range(n):
        trash.fill_(1)
        prep()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.gemm_tn(A, Bt, C, 0, tile=tile); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2] * 1e3
def write_a(): A.normal_()
def write_a_rows_interleaved():
    # u and a written together, chunk by chunk (like the fc1 epilogue): write-only kernels
    for i in range(0, M, 1576):
        U[i:i + 1576].normal_(); A[i:i + 1576].normal_()
def read_a(): A.sum()
print("cold                         %.1f us" % run(lambda: None))
print("A just written (normal_)       %.1f us" % run(write_a))
print("A+U written interleaved      %.1f us" % run(write_a_rows_interleaved))
print("A just read (sum)            %.1f us" % run(read_a))
print("warm back-to-back            %.1f us" % run(lambda: ops.gemm_tn(A, Bt, C, 0, tile=TILE)))
