"""End-to-end parity bars of the GPU model tests: per case (logits vs fp32 oracle, logits vs bf16-emulating oracle, worst parameter
gradient vs fp32 autograd), each set at ~1.5x the value measured on MI355X (round 2; the measured values are printed by the tests
and quoted in DESIGN.md section 2).  north_star's <= 1e-3 holds per kernel (tests/test_kernels_gpu.py); through a chain of bf16
roundings no bf16 evaluation - the reference's own included, which the bf16-emulating oracle restates - stays within it."""
import numpy as np

BARS = {}  # filled below: "family:case" -> (logits_vs_f32, logits_vs_bf16_emulation, worst_gradient)
GRAD_CAP = 2.5e-2  # no case may need more than this (was 6e-2 / 8e-2) ...
# ... except the 2 x 2 talking-heads matrices of the tiny CaiT (2 heads, 17 tokens): their gradient is a strongly cancelling sum of
# bf16-rounded scores x cotangents; measured 4.7e-2 at 3 images, 2.6e-2 at 32 (every other tensor of that model: <= 1.6e-2)
GRAD_CAP_EXEMPT = {"cait:tiny_cait:False": 4e-2, "cait:tiny_cait:True": 4e-2}


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def check_logits(tag, logits, ref32, refbf):
    r32, rbf, r_emul = rel(logits, ref32), rel(logits, refbf), rel(refbf, ref32)
    print(f"[{tag}] logits rel-L2: engine vs fp32 oracle {r32:.2e}, engine vs bf16-emulating oracle {rbf:.2e} "
          f"(bf16-emulating vs fp32 oracle {r_emul:.2e})")
    assert np.isfinite(logits).all()
    b32, bbf, _ = BARS[tag]
    assert r32 < b32, (tag, r32, b32)
    assert rbf < bbf, (tag, rbf, bbf)


def check_grads(tag, got, grads_ref, skip=()):
    worst, worst_k = 0.0, ""
    for k, g in grads_ref.items():
        if k in skip:
            continue
        assert got[k].shape == g.shape, k
        r = rel(got[k], g)
        if r > worst:
            worst, worst_k = r, k
    print(f"[{tag}] worst parameter-gradient rel-L2 vs fp32 autograd: {worst:.2e} ({worst_k})")
    bar = BARS[tag][2]
    assert bar <= GRAD_CAP_EXEMPT.get(tag, GRAD_CAP)
    assert worst < bar, (tag, worst_k, worst, bar)


# measured on MI355X in round 2 (gpurun_out/r2d/parity.log):  logits vs fp32 | logits vs bf16-emulation | worst gradient
BARS.update({
    "vit:tiny": (1.5e-2, 1.05e-2, 2.5e-2),       # 9.85e-3 | 6.82e-3 | 1.89e-2
    "vit:ti2": (1.4e-2, 1.25e-2, 2.4e-2),        # 9.20e-3 | 8.28e-3 | 1.57e-2
    "vit:s1_p32": (1.3e-2, 1.2e-2, 1.75e-2),     # 8.47e-3 | 7.99e-3 | 1.16e-2
    "vit:n577": (9.2e-3, 1.05e-2, 1.7e-2),       # 6.09e-3 | 6.89e-3 | 1.14e-2
    "vit:n1025": (9.2e-3, 1.05e-2, 1.7e-2),      # (round 5: the bars of n577, whose arithmetic the streaming kernels share; measured values in DESIGN.md)
    "vit:hd48": (1.3e-2, 1.2e-2, 2.5e-2),        # 8.77e-3 | 7.77e-3 | 1.80e-2
    "cait:tiny_cait:False": (1.35e-2, 1.45e-2, 4e-2),  # 8.73e-3 | 9.65e-3 | 2.60e-2 (32 images)
    "cait:tiny_cait:True": (1.35e-2, 1.45e-2, 4e-2),
    "cait:xxs2:True": (1.15e-2, 1.25e-2, 2.5e-2),  # 7.53e-3 | 8.24e-3 | 1.77e-2
    "cait:m1:True": (7e-3, 7.2e-3, 9.5e-3),        # 4.63e-3 | 4.75e-3 | 6.24e-3
    "mixer:tiny": (7.2e-3, 9.1e-3, 2.5e-2),      # 4.76e-3 | 6.05e-3 | 2.04e-2
    "mixer:p16": (5.6e-3, 6.1e-3, 1.5e-2),       # 3.73e-3 | 4.01e-3 | 9.75e-3
    "mixer:p32": (6.8e-3, 7.8e-3, 1.6e-2),       # 4.47e-3 | 5.15e-3 | 1.06e-2
    "tnt:tiny24": (1.2e-2, 1.65e-2, 2.5e-2),     # 7.77e-3 | 1.10e-2 | 2.15e-2
    "tnt:tiny40": (1.5e-2, 1.4e-2, 2.5e-2),      # 9.93e-3 | 9.27e-3 | 2.07e-2
    "tnt:b1": (1.15e-2, 1.25e-2, 2.5e-2),        # 7.52e-3 | 8.22e-3 | 1.84e-2
})
