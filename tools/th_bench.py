"""Times the materialising talking-heads attention (savit_th_attention_fwd / _bwd: th_scores, th_softmax, th_pv, th_bwd) at the CaiT-S24
layer geometry; under `rocprofv3 --kernel-trace --stats` the per-kernel split.  python tools/th_bench.py [B N H hd]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import savit_amd  # noqa: E402,F401
from savit_amd import lib  # noqa: E402

if os.environ.get("SAVIT_EXP_LIB"):
    lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "exp", "libsavit_%s.so" % os.environ["SAVIT_EXP_LIB"])
L = lib.load()
B, N, H, hd = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (256, 196, 8, 48)
d = H * hd
bf16 = torch.bfloat16
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * N, 3 * d, generator=g)
qkv[:, :d] *= hd ** -0.5
qkv = qkv.to(bf16).cuda()
d_o = torch.randn(B * N, d, generator=g).to(bf16).cuda()
T1 = (torch.eye(H) + 0.1 * torch.randn(H, H, generator=g)).cuda()
T2 = (torch.eye(H) + 0.1 * torch.randn(H, H, generator=g)).cuda()
o = torch.empty(B * N, d, dtype=bf16, device="cuda")
Np = (N + 7) // 8 * 8
sb, pb, dsb = (torch.empty(B, H, N, Np, dtype=bf16, device="cuda") for _ in range(3))
dqkv = torch.empty_like(qkv)
dT1, dT2 = torch.zeros(H, H, device="cuda"), torch.zeros(H, H, device="cuda")
ws = torch.empty(max(L.savit_th_attention_bwd_workspace_bytes(B, N, H), 16), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def fwd():
    return L.savit_th_attention_fwd(qkv.data_ptr(), T1.data_ptr(), T2.data_ptr(), sb.data_ptr(), pb.data_ptr(), o.data_ptr(), B, N, H, hd, 3 * d, Np, st)


def bwd():
    return L.savit_th_attention_bwd(qkv.data_ptr(), T1.data_ptr(), T2.data_ptr(), sb.data_ptr(), pb.data_ptr(), d_o.data_ptr(), dsb.data_ptr(),
                                    dqkv.data_ptr(), dT1.data_ptr(), dT2.data_ptr(), B, N, H, hd, 3 * d, Np, 1.0, ws.data_ptr(), ws.numel(), st)


def timeit(fn, n=20):
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def both():
    fwd()
    return bwd()


tf = timeit(fwd)
tfb = timeit(both)
print(f"B={B} N={N} H={H} hd={hd}: fwd {tf:.1f} us, fwd+bwd {tfb:.1f} us (bwd {tfb - tf:.1f})")
