"""Where does a 320x256 tile's time go?  (needs tools/pp320_stamps.patch applied to csrc/gemm_tn.hip and tools/build_variant.sh stamps gemm_tn.hip "-DPP320_STAMPS": per-wave s_memrealtime stamps at kernel entry, after the
prologue wait, after the main loop, after the last store has landed; 100 MHz constant clock)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import savit_amd
from savit_amd import lib as _l
_l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "exp", "libsavit_stamps.so")
from savit_amd import ops
bf16 = torch.bfloat16
M = 25216
for name, N, K in (("proj.dgrad", 768, 768), ("fc1.dgrad", 768, 3072), ("qkv", 2304, 768)):
    A = torch.randn(M, K, device="cuda").to(bf16); Bt = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf16)
    C = torch.empty(M, N, device="cuda", dtype=bf16)
    nwg = ((M + 319) // 320) * (N // 256)
    dbg = torch.zeros(nwg * 8 * 4 * 2, device="cuda", dtype=torch.float32)  # 8 waves x 4 stamps x 8 bytes
    junk = torch.empty(64 * 2 ** 20, device="cuda")
    for rep in range(4):
        junk.fill_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.gemm_tn(A, Bt, C, 0, tile=21, colsum=dbg); b.record(); torch.cuda.synchronize()
    st = dbg.view(torch.int64).cpu().numpy().reshape(nwg, 8, 4).astype(np.float64) / 100.0  # us
    t0 = st[:, :, 0].min()
    ent, pro, main, end = st[:, :, 0] - t0, st[:, :, 1] - st[:, :, 0], st[:, :, 2] - st[:, :, 1], st[:, :, 3] - st[:, :, 2]
    print(f"{name}: N={N} K={K} tiles={nwg} event {a.elapsed_time(b)*1e3:.1f} us | kernel span (first entry -> last store landed) {st[:,:,3].max()-t0:.1f} us")
    print(f"   WG entry after first WG: median {np.median(ent):.2f}  p90 {np.percentile(ent,90):.2f}  max {ent.max():.2f} us")
    print(f"   prologue (entry -> K-tile 0 landed + barrier): median {np.median(pro):.2f}  p90 {np.percentile(pro,90):.2f} us")
    print(f"   main loop: median {np.median(main):.2f}  p90 {np.percentile(main,90):.2f} us  ({K//64} K-tiles: {np.median(main)/(K//64):.3f} us each)")
    print(f"   epilogue (main loop end -> this wave's last store landed): median {np.median(end):.2f}  p90 {np.percentile(end,90):.2f}  max {end.max():.2f} us")
    print(f"   WG lifetime: median {np.median(st[:,:,3].max(1)-st[:,:,0].min(1)):.2f} us")
