"""Forensics on the LayerNorm-forward perturbation of round 1 (dev tool; GPU, no kernel of this repo is launched).

Round 1 logged wrong LayerNorm-forward outputs when two processes time-sliced one GPU (gpurun_out/k*_*.log).  This script
regenerates the probe's inputs (tools/kernel_det_probe.py: torch.manual_seed(0), same draw order) and asks, for every logged
event, which arithmetic reproduces the logged wrong value.  For a lane's element c of row i the kernel computes
y = (x - m) * r * g_c + b_c (m, r = row mean / rstd).  Hypotheses for the wrong value:
  H1  the in-place `v_pk_add_f32 x, x, -m` ran twice on the low half:           (x - 2 m) r g + b
  H2  the in-place multiply by g ran twice (narrow kernel only):                 (x - m) g g r + b
  H3  the in-place fma ran twice:                                                ((x - m) r g + b) r g + b  (wide) / ((x-m) g r + b) r + b (narrow)
  H4  the subtraction used the mean of another row (stale register): reported as the implied m' = m - dx, to be matched by eye
Output: per event the max abs error (in bf16 ulps of the logged value) of each hypothesis."""
import torch

EVENTS = [  # (tensor, first indices, cur, ref) copied from gpurun_out/kd_b.log, ks_a.log, ks_b.log, kh_a.log
    ("ln24", [3624, 3628, 3632, 3636, 3640, 3644], [2.5, -2.28125, -2.15625, -0.1708984375, -0.62890625, 1.2109375], [2.53125, -2.09375, -2.046875, -0.09765625, -0.671875, 1.265625]),
    ("ln384", [240962, 240966, 240970, 240974, 240978, 240982], [-2.34375, 1.3515625, -2.046875, 0.4921875, -0.8984375, 0.76953125], [-2.328125, 1.2421875, -2.09375, 0.462890625, -0.94140625, 0.8046875]),
    ("ln24", [554, 558, 562, 566, 570, 574], [-0.76953125, 1.015625, 0.1953125, 1.625, -1.9765625, -1.2734375], [-1.015625, 0.98046875, 0.46484375, 1.8046875, -1.75, -1.328125]),
    ("ln24", [296714, 296718, 296722, 296726, 296730, 296734], [-0.6953125, 0.6953125, -1.25, 2.546875, -0.5546875, -1.375], [-0.71875, 0.69140625, -1.21875, 2.5625, -0.53125, -1.3828125]),
    ("ln24", [1226, 1230, 1234, 1238, 1242, 1246], [0.5859375, 0.72265625, -0.07421875, 1.9765625, -1.5078125, -1.46875], [0.54296875, 0.71875, -0.02734375, 2.015625, -1.46875, -1.4765625]),
    ("ln24", [24456, 24460, 24464, 24468, 24472, 24476], [2.34375, 2.4375, -1.265625, -0.1904296875, 0.375, 0.62890625], [2.390625, 2.84375, -1.0234375, -0.039306640625, 0.283203125, 0.74609375]),
    ("ln24", [5256, 5260, 5264, 5268, 5272, 5276], [2.359375, -0.73046875, -0.93359375, -0.337890625, -0.1201171875, 0.486328125], [2.390625, -0.50390625, -0.796875, -0.25390625, -0.1708984375, 0.55078125]),
    ("ln24", [1128, 1132, 1136, 1140, 1144, 1148], [2.671875, 0.73828125, -0.8828125, -0.59375, -0.0284423828125, 0.80859375], [2.640625, 0.466796875, -1.0390625, -0.6953125, 0.032958984375, 0.7265625]),
    ("ln384", [152646, 152650, 152654, 152658, 152662, 152666], [1.9296875, -0.05712890625, 0.671875, -0.41796875, 0.46875, 0.8046875], [1.9140625, -0.062255859375, 0.66796875, -0.421875, 0.47265625, 0.80859375]),
    ("ln384", [223682, 223686, 223690, 223694, 223698, 223702], [-1.6953125, 0.0235595703125, -0.25390625, -0.7421875, -0.259765625, 1.2578125], [-1.7109375, 0.138671875, -0.2021484375, -0.7109375, -0.21875, 1.2265625]),
    ("ln24", [5928, 5932, 5936, 5940, 5944, 5948], [2.46875, 2.3125, -0.267578125, 0.421875, -0.26953125, 1.0234375], [2.390625, 1.640625, -0.65625, 0.177734375, -0.12109375, 0.83203125]),
    ("ln24", [2186, 2190, 2194, 2198, 2202, 2206], [-0.7109375, 1.234375, 0.671875, 2.375, -1.2734375, -1.359375], [-1.0703125, 1.1875, 1.0625, 2.640625, -0.953125, -1.4375]),
    ("ln24", [6314, 6318, 6322, 6326, 6330, 6334], [-0.337890625, 0.80078125, 0.2158203125, 2.546875, -0.341796875, -1.75], [-0.5625, 0.76953125, 0.462890625, 2.71875, -0.1357421875, -1.796875]),
]


def main():
    dev = "cuda"
    torch.manual_seed(0)
    rows = 4 * 196 * 16
    x24 = torch.randn(rows, 24, device=dev); g24 = torch.randn(24, device=dev); b24 = torch.randn(24, device=dev)
    x384 = torch.randn(4 * 197, 384, device=dev); g384 = torch.randn(384, device=dev); b384 = torch.randn(384, device=dev)
    T = {"ln24": (x24.double().cpu(), g24.double().cpu(), b24.double().cpu()), "ln384": (x384.double().cpu(), g384.double().cpu(), b384.double().cpu())}
    bf = lambda t: t.float().to(torch.bfloat16).double()
    for name, idx, cur, ref in EVENTS:
        x, g, b = T[name]
        d = x.shape[1]
        narrow = d == 24
        cur = torch.tensor(cur, dtype=torch.float64); ref = torch.tensor(ref, dtype=torch.float64)
        i = torch.tensor(idx) // d; c = torch.tensor(idx) % d
        xv = x[i, c]; m = x[i].mean(1); var = (x[i] ** 2).mean(1) - m * m; r = (var + 1e-6).rsqrt()
        gc, bc = g[c], b[c]
        y0 = (xv - m) * r * gc + bc
        ulp = lambda v: torch.maximum(v.abs(), torch.tensor(2.0 ** -126, dtype=torch.float64)).log2().floor().exp2() * 2.0 ** -7
        err = lambda h: float(((bf(h) - cur).abs() / ulp(cur)).max())
        h1 = (xv - 2 * m) * r * gc + bc
        h2 = (xv - m) * gc * gc * r + bc
        h3 = (((xv - m) * gc * r + bc) * r + bc) if narrow else (y0 * r * gc + bc)
        dx = (cur - ref) / (r * gc)  # implied shift of x
        print(f"{name} row {int(i[0])} cols {c.tolist()}  ref-check {err(y0) if False else float(((bf(y0) - ref).abs() / ulp(ref)).max()):.1f} ulp | H1 {err(h1):.1f}  H2 {err(h2):.1f}  H3 {err(h3):.1f} ulp"
              f" | row mean {float(m[0]):+.4f} implied dx {[round(float(v), 3) for v in dx]}")


if __name__ == "__main__":
    main()
