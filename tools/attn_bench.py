"""Times savit_attention_fwd / savit_attention_bwd at a layer geometry (default DeiT-B: 128 images, 12 heads of 64, N = 197).
python tools/attn_bench.py [B N H hd]"""
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
B, N, H, hd = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (128, 197, 12, 64)
d = H * hd
bf16 = torch.bfloat16
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * N, 3 * d, generator=g)
qkv[:, :d] *= hd ** -0.5
qkv = qkv.to(bf16).cuda()
d_o = torch.randn(B * N, d, generator=g).to(bf16).cuda()
o = torch.empty(B * N, d, dtype=bf16, device="cuda")
lse = torch.empty(B, H, N, device="cuda")
dqkv = torch.empty_like(qkv)
st = torch.cuda.current_stream().cuda_stream


def fwd():
    return L.savit_attention_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, H, hd, 3 * d, st)


def bwd():
    return L.savit_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, N, H, hd, 3 * d, 1.0, st)


def timeit(fn, n=50):
    for _ in range(5):
        assert fn() == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


bytes_f = B * N * (3 * d + d) * 2
bytes_b = B * N * (3 * d + d + d + 3 * d) * 2
tf, tb = timeit(fwd), timeit(bwd)
print(f"B={B} N={N} H={H} hd={hd}: fwd {tf:.1f} us ({bytes_f / tf / 1e6:.2f} TB/s), "
      f"bwd {tb:.1f} us ({bytes_b / tb / 1e6:.2f} TB/s)")
