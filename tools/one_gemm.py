import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import ops
bf16 = torch.bfloat16
M, N, K, tile = [int(x) for x in sys.argv[1:5]]
A = torch.randn(M, K, device="cuda").to(bf16); Bt = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf16)
C = torch.empty(M, N, device="cuda", dtype=bf16)
for _ in range(5):
    ops.gemm_tn(A, Bt, C, 0, tile=tile)
torch.cuda.synchronize()
