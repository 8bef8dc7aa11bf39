"""Repeat-launch stress of the persistent attention kernels: 60 backward + forward launches per geometry must agree bit for bit (dev tool)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import savit_amd
from savit_amd import lib
L = lib.load()
bf16 = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
bad = 0
for (B, N, H) in [(128, 197, 12), (37, 197, 12), (256, 197, 6), (65, 64, 12), (300, 33, 4), (40, 224, 5), (9, 161, 7), (1000, 17, 2)]:
    d = H * 64
    qkv = (torch.randn(B * N, 3 * d, device="cuda") * 0.7).to(bf16)
    d_o = torch.randn(B * N, d, device="cuda").to(bf16)
    o = torch.empty(B * N, d, dtype=bf16, device="cuda")
    lse = torch.empty(B, H, N, device="cuda")
    assert L.savit_attention_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, H, 64, 3 * d, st) == 0
    o0 = o.clone(); lse0 = lse.clone()
    ref = None
    for it in range(60):
        dq = torch.full_like(qkv, float("nan"))
        assert L.savit_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dq.data_ptr(), B, N, H, 64, 3 * d, 1.0, st) == 0
        o2 = torch.empty_like(o); l2 = torch.empty_like(lse)
        assert L.savit_attention_fwd(qkv.data_ptr(), o2.data_ptr(), l2.data_ptr(), B, N, H, 64, 3 * d, st) == 0
        torch.cuda.synchronize()
        if ref is None:
            ref = dq.clone()
            assert torch.isfinite(ref.float()).all(), "non-finite"
        elif not torch.equal(dq, ref):
            bad += 1
            print("MISMATCH bwd", B, N, H, it, (dq.float() - ref.float()).abs().max().item())
        if not (torch.equal(o2, o0) and torch.equal(l2, lse0)):
            bad += 1
            print("MISMATCH fwd", B, N, H, it)
    print("ok", B, N, H)
print("bad", bad)
