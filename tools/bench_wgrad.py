import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import ops
bf16 = torch.bfloat16
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
B, d = 128, 768
M, F = B * 197, 4 * d
flush = torch.empty(512 * 2 ** 20, dtype=torch.uint8, device="cuda")
for name, Kin, Nout in (("Wqkv", d, 3 * d), ("Wo", d, d), ("W1", d, F), ("W2", F, d)):
    X = torch.randn(M, Kin, device="cuda").to(bf16); dY = torch.randn(M, Nout, device="cuda").to(bf16)
    dW = torch.zeros(Kin, Nout, device="cuda")
    fl = 2.0 * M * Kin * Nout
    def base():
        flush.zero_()
    tb = timeit(base)
    res = []
    for sp in (0, 1, 2, 3, 4, 5, 7, 9):
        def run():
            flush.zero_(); ops.gemm_wgrad(X, dY, dW, splits=sp)
        t = timeit(run) - tb
        res.append(f"s{sp}:{t*1e6:6.1f}us/{fl/t/1e12:4.0f}")
    print(f"variant {os.environ.get('SAVIT_WGRAD_VARIANT','0')} {name:5s}: " + " ".join(res))
