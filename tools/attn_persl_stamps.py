"""Where does an item of the loader-wave attention backward spend its time?  (round 6; needs tools/attn_persl_stamps.patch applied to
csrc/attention.hip and tools/build_variant.sh lstamp attention.hip "": per-wave s_memrealtime stamps behind the LSE buffer)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import savit_amd
from savit_amd import lib as _l
_l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "exp", "libsavit_lstamp.so")
L = _l.load()
bf16 = torch.bfloat16
B, N, H, hd, NT = 128, 197, 12, 64, 7
d = H * hd
qkv = (torch.randn(B * N, 3 * d, device="cuda") * 0.5).to(bf16)
o = torch.randn(B * N, d, device="cuda").to(bf16)
do = torch.randn(B * N, d, device="cuda").to(bf16)
nitems = B * H
lse_dbg = torch.zeros(B * H * N + nitems * NT * 8 * 2 + 64, device="cuda")
lse_dbg[:B * H * N] = torch.randn(B * H * N, device="cuda").abs() + 3.0
dqkv = torch.empty_like(qkv)
s = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    rc = L.savit_attention_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse_dbg.data_ptr(), dqkv.data_ptr(), B, N, H, hd, 3 * d, 0.125, s)
    b.record(); torch.cuda.synchronize()
    assert rc == 0
off = B * H * N
st = lse_dbg[off:off + nitems * NT * 8 * 2].view(torch.int64).cpu().numpy().reshape(nitems, NT, 8).astype(np.float64) / 100.0
print(f"loader-wave attention backward B={B} N={N} H={H}: event {a.elapsed_time(b)*1e3:.1f} us, span {st[:,:,7].max()-st[:,:,0].min():.1f} us")
names = ["pass B", "own rows + next K/V row requests + dK,dV stores (issue)", "wait at barrier b", "pass A", "wait at barrier c",
         "vmcnt(0) + delta + rows->fragments + dQ store (issue)", "wait at barrier a"]
for i, nm in enumerate(names):
    dur = st[:, :, i + 1] - st[:, :, i]
    print(f"  {nm:58s} median {np.median(dur):6.2f} us  p10 {np.percentile(dur,10):6.2f}  p90 {np.percentile(dur,90):6.2f}  per wave: " + " ".join(f"{np.median(dur[:,w]):5.2f}" for w in range(NT)))
life = st[:, :, 7].max(1) - st[:, :, 0].min(1)
print(f"  item (barrier a to barrier a) median {np.median(life):.2f} us")
