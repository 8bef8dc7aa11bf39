"""Where does attention backward spend its time?  (needs tools/attn_stamps.patch applied to csrc/attention.hip and tools/build_variant.sh astamp attention.hip "-DATTN_STAMPS": per-wave s_memrealtime stamps)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import savit_amd
from savit_amd import lib as _l
_l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "exp", "libsavit_%s.so" % os.environ.get("STAMP_LIB", "astamp") + "")
L = _l.load()
bf16 = torch.bfloat16
B, N, H, hd = 128, 197, 12, 64
d = H * hd
NT = 7
qkv = (torch.randn(B * N, 3 * d, device="cuda") * 0.5).to(bf16)
o = torch.randn(B * N, d, device="cuda").to(bf16)
do = torch.randn(B * N, d, device="cuda").to(bf16)
nitems = B * H
lse_dbg = torch.zeros(B * H * N + nitems * NT * 8 * 2 + 64, device="cuda")  # LSE + stamps (8 bytes each)
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
if off % 2: off += 0
st = lse_dbg[off:off + nitems * NT * 5 * 2].view(torch.int64).cpu().numpy().reshape(nitems, NT, 5).astype(np.float64) / 100.0
t0 = st[:, :, 0].min()
print(f"attention backward B={B} N={N} H={H}: event {a.elapsed_time(b)*1e3:.1f} us, span {st[:,:,4].max()-t0:.1f} us, items {nitems}")
names = ["stage (entry -> images landed, delta)", "pass A (dQ)", "dQ store issue + barrier + pass B", "dK,dV stores landed"]
if os.environ.get("PERS"): names = ["DMA K,V issue + pass B", "own Q,dO rows + barrier b + dK,dV stores + row requests + DMA next Q,dO (issue)", "pass A", "delta(next) + barrier a"]
for i, nm in enumerate(names):
    dur = st[:, :, i + 1] - st[:, :, i]
    print(f"  {nm:42s} median {np.median(dur):6.2f} us  p10 {np.percentile(dur,10):6.2f}  p90 {np.percentile(dur,90):6.2f}  per wave: " + " ".join(f"{np.median(dur[:,w]):5.2f}" for w in range(NT)))
life = st[:, :, 4].max(1) - st[:, :, 0].min(1)
print(f"  WG lifetime median {np.median(life):.2f} us; entry times of the k-th generation of WGs: ", [round(float(np.median(np.sort(st[:,0,0]-t0)[i*256:(i+1)*256])),1) for i in range(6)])

if os.environ.get("PERS"):
    sx = lse_dbg[off + nitems * NT * 5 * 2: off + nitems * NT * 8 * 2].view(torch.int64).cpu().numpy().reshape(nitems, NT, 3).astype(np.float64) / 100.0
    ok = sx[:, :, 2] > 0
    for nm, d in (("  .. own rows + barrier b", sx[:, :, 0] - st[:, :, 1]), ("  .. dK,dV stores issue", sx[:, :, 1] - sx[:, :, 0]), ("  .. row requests issue", sx[:, :, 2] - sx[:, :, 1]), ("  .. DMA issue", st[:, :, 2] - sx[:, :, 2])):
        d = np.where(ok, d, np.nan)
        print(f"{nm:30s} median {np.nanmedian(d):6.2f}  per wave: " + " ".join(f"{np.nanmedian(d[:, w]):5.2f}" for w in range(NT)))
