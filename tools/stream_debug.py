"""Debug aid for the streaming GEMM (tile 30): where does it differ from tile 13?  python tools/stream_debug.py epi M N K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import savit_amd
from savit_amd import ops
bf16 = torch.bfloat16
epi, M, N, K = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (3, 19700, 1536, 768)))
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.randn(M, K, device="cuda", generator=g).to(bf16)
Bt = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(bf16)
bias = 0.1 * torch.randn(N, device="cuda", generator=g)
u = torch.randn(M, N, device="cuda", generator=g).to(bf16)
def run(tile):
    C = torch.full((M, N), float("nan"), device="cuda", dtype=bf16)
    kw = {}
    if epi == 0: kw = dict(alpha=0.125, alpha_cols=N // 3 // 128 * 128)
    if epi == 1: kw = dict(bias=bias, C2=torch.full((M, N), float("nan"), device="cuda", dtype=bf16))
    if epi == 3: kw = dict(aux=u, colsum=torch.full((ops.gemm_colsum_rows(M, N, K, tile), N), float("nan"), device="cuda"))
    ops.gemm_tn(A, Bt, C, epi, tile=tile, **kw)
    torch.cuda.synchronize()
    return [C] + [kw[k] for k in ("C2", "colsum") if k in kw]
a, b = run(13), run(30)
for i, (x, y) in enumerate(zip(a, b)):
    nan = torch.isnan(y.float())
    print(f"out{i}: shape {tuple(y.shape)} NaN {int(nan.sum())}")
    if int(nan.sum()):
        r, c = torch.nonzero(nan, as_tuple=True)
        print("   NaN rows", sorted(set((r // 8 * 8).tolist()))[:20], "... cols", sorted(set((c // 64 * 64).tolist()))[:30])
    if x.shape == y.shape:
        d = (x.float() != y.float()) & ~nan
        print(f"   differing (non-NaN) elements vs tile 13: {int(d.sum())}")
        if int(d.sum()):
            r, c = torch.nonzero(d, as_tuple=True)
            print("   rows//8", sorted(set((r // 8).tolist()))[:30], "cols//64", sorted(set((c // 64).tolist()))[:30])
    else:
        print("   column sums rel diff", float((x.sum(0) - y.sum(0)).norm() / x.sum(0).norm()))
# per-tile map of corrupted elements (256 x 128 tiles)
x, y = a[0].float(), b[0].float()
bad = (x != y) | torch.isnan(y)
tm_n, tn_n = (M + 255) // 256, N // 128
pad = torch.zeros(tm_n * 256, N, dtype=torch.bool, device="cuda")
pad[:M] = bad
cnt = pad.view(tm_n, 256, tn_n, 128).sum(dim=(1, 3)).cpu()
print("tiles with corrupted elements:", int((cnt > 0).sum()), "of", tm_n * tn_n)
for tm in range(tm_n):
    row = "".join("#" if int(cnt[tm, tn]) else "." for tn in range(tn_n))
    if "#" in row:
        print(f"tm {tm:3d} {row}")
if epi == 3:
    xs, ys = a[1].view(tm_n, -1, N).sum(1), b[1].view(tm_n, -1, N).sum(1)
    d = (xs - ys).abs()
    print("column-sum slab, per row tile: max |diff| per tile", [round(float(v), 4) for v in d.max(dim=1).values[:16]], "...")
    worst = int(d.max(dim=1).values.argmax())
    wc = d[worst].topk(8)
    print(f"worst row tile {worst}: columns {wc.indices.tolist()} diffs {[round(float(v), 4) for v in wc.values]}  ref {[round(float(xs[worst, c]), 3) for c in wc.indices]}")
    # which wave rows of tile 30's slab hold the difference: recompute the exact per-(64-row block) sums from the output tensor
    blk = torch.zeros(tm_n * 256, N, device="cuda"); blk[:M] = b[0].float()
    exact = blk.view(tm_n * 4, 64, N).sum(1)
    dd = (exact - b[1]).abs()
    print("tile-30 slab vs sums recomputed from its own output: max diff", float(dd.max()), "rows with diff > 1e-2:", (dd.max(dim=1).values > 1e-2).nonzero().flatten()[:24].tolist())
    cols = (dd > 1e-2).any(dim=0).nonzero().flatten()
    print("columns with diff > 1e-2 (mod 64):", sorted(set((cols % 64).tolist()))[:64])
