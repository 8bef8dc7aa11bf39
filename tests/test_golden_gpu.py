"""GPU parity against the COMMITTED fixtures of tests/golden/ (SURVEY 8c "Golden vectors").

Two sets: single encoder blocks at the real widths of the BASELINE configs (block_*.npz: d 192 / 384 / 768 / 1024, N 197 / 577, the
CaiT talking-heads + class-attention block) and small two-layer models of all four families (e2e_*.npz).  Parameters and images
are regenerated from the fixture's seed and pinned by the stored checksums; expected logits (fp64 and bf16-emulated), loss and
gradients (norm + 2048 sampled entries per tensor) are stored.  No oracle forward runs here: the stored numbers are the judge.
(The d = 32 tiny_*.npz fixtures are narrower than the engines' minimum width; tests/test_oracle.py uses them to pin the oracle.)

Round 3: every fixture also stores `logits_engine`, the oracle evaluated with the ENGINE's rounding points (vit_ref.Policy("engine"):
one rounding per fused GEMM epilogue, fp32 scores and softmax statistics, bf16 P only as the MFMA operand).  Against that array the
engine's distance is summation order plus rounding-boundary flips - a bar that a small kernel error would break, which the two
policy-distance bars above it cannot (missing 5).

Bars = ~1.5x the values measured on MI355X (printed by the tests):
  logits vs fp64 oracle            7.4e-3 - 9.7e-3 measured on the single blocks (the bf16-emulating oracle itself: 8.8e-3 - 1.0e-2)
  logits vs bf16-emulating oracle  7.6e-3 - 8.6e-3 measured: two bf16 evaluations that round at the same places but sum in different
                                   orders differ by as much as either differs from exact math
  gradients vs fp64 autograd       norm within 1.5e-2, sampled entries rel-L2 <= 2.5e-2
north_star's "<= 1e-3 rel bf16" holds per kernel (tests/test_kernels_gpu.py: each kernel vs exact math on the same bf16 inputs)
and cannot hold for a chain of bf16 roundings: DESIGN.md section 2 has the per-layer growth curve."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref, vit_ref
from tests.golden import make_golden

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _engine(kw, B, **opts):
    from savit_amd.config import ModelConfig

    mc = ModelConfig(**kw)
    if mc.kind == "cait":
        from savit_amd.cait_engine import CaiTEngine

        return CaiTEngine(mc, B)
    if mc.kind == "mixer":
        from savit_amd.mixer_engine import MixerEngine

        return MixerEngine(mc, B)
    if mc.kind == "tnt":
        from savit_amd.tnt_engine import TNTEngine

        return TNTEngine(mc, B)
    from savit_amd.engine import ViTEngine

    return ViTEngine(mc, B, **opts)


def _flat(tree):
    return {k: v.detach().float().cpu().numpy() for k, v in torch_ref.leaves(tree)}


# name: (logits vs fp64, logits vs bf16-emulation, gradient norm, gradient samples) = ~1.5x the values measured on MI355X
BLOCK_BARS = {  # measured (gpurun_out/r2d/parity.log):          logits f64 | bf16-emu | grad norm | grad samples
    "block_d192_n197": (1.45e-2, 1.25e-2, 5e-3, 2.5e-2),         # 9.66e-3 | 8.07e-3 | 3.2e-3 | 2.08e-2
    "block_d384_n197": (1.25e-2, 1.2e-2, 5e-3, 1.8e-2),          # 8.29e-3 | 7.90e-3 | 2.3e-3 | 1.15e-2
    "block_d768_n197": (1.3e-2, 1.2e-2, 5e-3, 1.8e-2),           # 8.73e-3 | 7.68e-3 | 1.9e-3 | 1.20e-2
    "block_d1024_n577": (1.25e-2, 1.15e-2, 5e-3, 1.7e-2),        # 8.22e-3 | 7.59e-3 | 1.1e-3 | 1.10e-2
    "block_cait_d384_n196": (1.15e-2, 1.3e-2, 5e-3, 2.2e-2),     # 7.44e-3 | 8.64e-3 | 1.8e-3 | 1.42e-2
    "e2e_vit_d128": (1.25e-2, 1.2e-2, 6e-3, 2.1e-2),             # 8.20e-3 | 7.69e-3 | 3.4e-3 | 1.38e-2
    "e2e_cait_d128": (1e-2, 1.35e-2, 1.4e-2, 2.5e-2),            # 6.42e-3 | 8.74e-3 | 9.0e-3 | 1.79e-2
    "e2e_mixer_d128": (1.15e-2, 1.25e-2, 1.5e-2, 2.5e-2),        # 7.55e-3 | 8.15e-3
    "e2e_tnt_d128": (1.4e-2, 1.85e-2, 1.8e-2, 2.5e-2),           # 9.23e-3 | 1.23e-2 | 1.16e-2 | 2.34e-2
    # round 6: two-layer real-width models (layer 0 = the dense block, layer 1 = the cls-row path by default); measured values
    # are printed by the test and quoted next to the bars (profiles/r06_parity_dense_and_default.log)
    # (default plan | dense plan where it differs)
    "block2_d768_n197": (1.35e-2, 1.26e-2, 5e-3, 2.3e-2),        # 8.93e-3 | 8.39e-3 | 1.5e-3 | 1.51e-2
    "block2_d1024_n577": (1.2e-2, 1.2e-2, 5e-3, 2.3e-2),         # 7.84e-3 (dense 7.77e-3) | 7.83e-3 (7.85e-3) | 2.0e-3 (2.3e-3) | 1.52e-2 (1.47e-2)
    "block2_d384_n197": (1.27e-2, 1.4e-2, 5e-3, 2.5e-2),         # 8.42e-3 | 9.17e-3 | 1.9e-3 | 1.84e-2 (the head kernel; bar capped at parity_bars.GRAD_CAP)
}


# engine vs the oracle evaluated with the engine's own rounding points (`logits_engine`): ~1.5x measured (round 3; MI355X:
# d768 4.48e-3, d1024 5.58e-3, CaiT block 6.26e-3, e2e CaiT 4.96e-3, the rest < 4e-3).  These are NOT at the "summation order" level
# (1e-6) the policy was built to expose, and cannot be: evaluating this very policy on the CPU with fp64 instead of fp32 accumulation -
# identical rounding points, only the summation precision differs - moves the logits of the same blocks by 3.4e-3 (d 192) and 4.7e-3
# (d 768).  A 1e-6 difference flips one bf16 rounding in 4 000; every flip is a 4e-3 perturbation of that element, which flips 1 in
# 100 of the next tensor's roundings, and so on: within the half dozen rounding stages of ONE encoder block any two evaluations
# decorrelate to the bf16 noise floor.  Parity below that floor is a per-kernel property (tests/test_kernels_gpu.py: each kernel
# against exact math with its own operand roundings emulated, 1e-4 ... 1e-3).
ENGINE_POLICY_BARS = {
    "block_d192_n197": 4e-3, "block_d384_n197": 4e-3, "block_d768_n197": 6.8e-3, "block_d1024_n577": 8.4e-3,
    "block_cait_d384_n196": 9.4e-3, "e2e_vit_d128": 4e-3, "e2e_cait_d128": 7.5e-3, "e2e_mixer_d128": 1e-2, "e2e_tnt_d128": 1.5e-2,
    "block2_d768_n197": 8.6e-3, "block2_d1024_n577": 8.8e-3, "block2_d384_n197": 9.4e-3,  # round 6, MI355X: 5.73e-3, 5.83e-3 (dense 5.77e-3), 6.25e-3
}


def _cases():
    """(fixture, plan): every ViT-family fixture runs twice - the product default (last layer on the cls rows: engine.cls_only_last /
    cls_fwd) and the DENSE plan (`cls_only_last=False`: every token of every layer through dense attention, the dense proj / fc1-GELU /
    fc2-residual / GELU' epilogues and the dense LayerNorm backward).  Round 5 made the cls-row path the default, which took the dense
    block of the one-layer fixtures - incl. the N = 577 general attention kernels of block_d1024_n577 - off the oracle-compared path
    (VERDICT r5 weak 1); the dense runs put it back, the two-layer block2_* fixtures cover both paths in one model."""
    out = []
    for name in sorted(make_golden.BLOCKS):
        out.append((name, "default"))
        if make_golden.BLOCKS[name][0]["kind"] == "vit":
            out.append((name, "dense"))
    return out


@pytest.mark.parametrize("name,plan", _cases())
def test_real_width_block_fixture(name, plan):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    kw, seed = make_golden.BLOCKS[name]
    assert json.loads(str(fx["cfg"])) == kw and int(fx["seed"]) == seed
    cfg, params, images, labels = make_golden.block_inputs(kw, seed)
    for k, v in make_golden.checksums(params, images).items():  # the regenerated inputs ARE the fixture's inputs
        np.testing.assert_allclose(v, fx[k], rtol=1e-12, atol=1e-12, err_msg=k)
    assert np.array_equal(labels, fx["labels"])
    B = images.shape[0]
    eng = _engine(kw, B, **({"cls_only_last": False} if plan == "dense" else {}))
    if cfg.kind == "vit":
        assert eng.cls_only_last == (plan == "default") and (plan == "default" or not eng.cls_fwd)
    name = f"{name}:{plan}" if plan != "default" else name
    eng.load_params(params)
    x = torch.as_tensor(images).cuda()
    logits = (eng.forward(x, is_training=False) if cfg.kind == "cait" else eng.forward(x)).float().cpu().numpy()
    if cfg.kind == "vit":
        # how many layers push every token through dense attention in this run (the oracle comparison below covers exactly those)
        dense_attn = sum(1 for c in eng._fwd_plan.calls if c[0] is eng.L.savit_attention_fwd)
        assert dense_attn == (cfg.num_layers if plan == "dense" else cfg.num_layers - (1 if eng.cls_fwd else 0)), dense_attn
    base = name.split(":")[0]
    b64, bbf, bgn, bgs = BLOCK_BARS[base]
    r64, rbf, r_emul = rel(logits, fx["logits"]), rel(logits, fx["logits_bf16"]), rel(fx["logits_bf16"], fx["logits"])
    reng = rel(logits, fx["logits_engine"])
    print(f"[{name}] logits rel-L2: engine vs fp64 {r64:.2e}, engine vs bf16-emulation {rbf:.2e} (bf16-emulation vs fp64 {r_emul:.2e}), "
          f"engine vs engine-policy oracle {reng:.2e}")
    assert np.isfinite(logits).all()
    bad = []  # every figure is printed before the first failure is raised
    if not (r64 < b64 and rbf < bbf):
        bad.append(("logits", r64, rbf))
    if not reng < ENGINE_POLICY_BARS[base]:
        bad.append(("logits vs engine-policy oracle", reng, ENGINE_POLICY_BARS[base]))
    loss = float(eng.loss_backward(torch.as_tensor(labels).cuda(), 0.1))
    if not abs(loss - float(fx["loss"])) < 5e-3 * max(1.0, abs(float(fx["loss"]))):
        bad.append(("loss", loss, float(fx["loss"])))
    got = _flat(eng.grad_tree()["params"])
    keys = [k[3:] for k in fx.files if k.startswith("GN:")]
    assert set(keys) == set(got)
    worst_n, worst_s = (0.0, ""), (0.0, "")
    gn_max = max(float(fx["GN:" + k]) for k in keys)
    for k in keys:
        g = got[k].astype(np.float64).ravel()
        gn = float(fx["GN:" + k])
        if gn < 1e-9 * gn_max:
            # a gradient that is exactly zero in exact arithmetic (the Mixer's token-Dense output bias: every later LayerNorm removes the
            # shift it causes; tests/test_mixer_oracle.py) - both sides hold rounding noise only: bounded against its sibling bias
            scale = float(fx["GN:" + k.replace("Dense_1/bias", "Dense_0/bias")])
            if not np.linalg.norm(g) < 2e-2 * scale:
                bad.append((k, float(np.linalg.norm(g)), scale))
            continue
        rn = abs(np.linalg.norm(g) - gn) / gn
        rs = rel(g[fx["GI:" + k]], fx["GV:" + k]) if gn > 0 else 0.0
        worst_n = max(worst_n, (rn, k))
        worst_s = max(worst_s, (rs, k))
        if not (rn < bgn and rs < bgs):
            bad.append((k, rn, rs))
    print(f"[{name}] gradients vs fp64 autograd: worst norm deviation {worst_n[0]:.2e} ({worst_n[1]}), worst sampled rel-L2 {worst_s[0]:.2e} ({worst_s[1]})")
    assert not bad, bad
