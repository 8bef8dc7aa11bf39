"""Full BASELINE-size checks (DeiT-B/16 at 128 images, ViT-L/16 at 384^2): the CPU oracle cannot run these sizes in test time, so
parity is asserted through size-independent properties of the path (task section 3): the zero-init-head known answers of the
reference (SURVEY 8c i), per-sample independence (a batch row never sees another row: no BatchNorm on this path),
permutation equivariance, linearity of the mean gradient over sub-batches (what data parallelism relies on), repeatability,
and a descending loss on a fixed batch."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def deit_b():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import savit_amd  # noqa: F401
    from savit_amd.config import get_config
    from savit_amd.engine import ViTEngine

    cfg = get_config("vit_b_patch16")
    eng = ViTEngine(cfg, 128)
    eng.init_params(42)
    g = torch.Generator(device="cuda").manual_seed(1)
    img = torch.randn(128, 224, 224, 3, device="cuda", generator=g).to(torch.bfloat16)
    lab = torch.randint(0, 1000, (128,), device="cuda", generator=g, dtype=torch.int32)
    return cfg, eng, img, lab


def _assert_backward_repeats(eng, g_a, g_b):
    """Two backward passes over the same batch: every Dense-kernel gradient (the weight-gradient GEMMs: fixed-order sums, no atomics
    since round 2) must repeat BITWISE; bias / LayerNorm / embedding gradients end in a handful of fp32 atomics per element (column
    sums folded by 8 blocks, the loss kernel's bias gradient) and repeat to summation order."""
    checked = 0
    for name, (off, shape) in eng.layout.off.items():
        n = 1
        for s_ in shape:
            n *= s_
        a, b = g_a[off:off + n], g_b[off:off + n]
        if len(shape) == 2 and name.split(".")[-1] in ("Wqkv", "Wo", "W1", "W2", "Wh", "Wpe"):
            assert torch.equal(a, b), f"{name}: weight gradient not bitwise reproducible"
            checked += 1
        elif float(b.norm()) > 0:
            assert float((a - b).norm() / b.norm()) < 1e-5, name
    assert checked >= 4 * eng.cfg.num_layers


def _head(eng, cfg, seed=7):
    g = torch.Generator().manual_seed(seed)
    eng.layout.view(eng.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes, generator=g) * cfg.embed_dim ** -0.5)
    eng.weights_stale = True


def test_zero_head_known_answers_full_size(deit_b):
    cfg, eng, img, lab = deit_b
    eng.init_params(42)  # reference initialisers: zero head kernel and bias (vit.py:96-98)
    logits = eng.forward(img)
    assert float(logits.abs().max()) == 0.0
    loss = eng.loss_backward(lab, label_smoothing=0.1)
    torch.cuda.synchronize()
    assert abs(float(loss) - math.log(1000.0)) < 1e-4  # train.py:83-90 with uniform softmax
    gt = eng.layout.flax_tree(eng.grads)["params"]
    assert float(gt["Dense_0"]["bias"].abs().max()) > 0  # dL/dbias = 1/1000 - y_smooth
    assert float(gt["Dense_0"]["kernel"].abs().max()) > 0
    enc = gt["Encoder_0"]["EncoderBlock_5"]
    assert float(enc["FFBlock_0"]["Dense_0"]["kernel"].abs().max()) == 0.0  # nothing flows past a zero head kernel
    assert float(gt["PatchEmbedBlock_0"]["Dense_0"]["kernel"].abs().max()) == 0.0


def test_rows_are_independent_and_permutation_equivariant(deit_b):
    cfg, eng, img, lab = deit_b
    eng.init_params(42)
    _head(eng, cfg)
    full = eng.forward(img).clone()
    assert torch.isfinite(full).all() and float(full.abs().max()) > 0.1
    # repeatability of the forward pass: bitwise
    assert torch.equal(eng.forward(img), full)
    # permuting the images permutes the logits (bitwise: every row runs the same instruction sequence)
    perm = torch.randperm(128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    assert torch.equal(eng.forward(img[perm].contiguous()), full[perm])
    # a 16-image engine (other GEMM grids, same K order) reproduces the rows of the 128-image batch
    from savit_amd.engine import ViTEngine

    small = ViTEngine(cfg, 16)
    small.params, small.w, small.weights_stale = eng.params, eng.w, False
    sub = small.forward(img[32:48].contiguous())
    err = float((sub - full[32:48]).norm() / full[32:48].norm())
    assert err < 1e-6, err  # measured 0: every output element sums its K products in the same order whatever the grid


def test_mean_gradient_is_linear_over_sub_batches(deit_b):
    """grad(mean loss over 128) == mean of the two 64-image gradients: the identity data parallelism rests on (train.py:96)."""
    cfg, eng, img, lab = deit_b
    eng.init_params(42)
    _head(eng, cfg)
    eng.forward(img)
    eng.loss_backward(lab, label_smoothing=0.1)
    torch.cuda.synchronize()
    g_full = eng.grads.clone()
    from savit_amd.engine import ViTEngine

    half = ViTEngine(cfg, 64)
    half.params, half.w, half.weights_stale = eng.params, eng.w, False
    acc = torch.zeros_like(g_full)
    for lo in (0, 64):
        half.forward(img[lo:lo + 64].contiguous())
        half.loss_backward(lab[lo:lo + 64].contiguous(), label_smoothing=0.1)
        torch.cuda.synchronize()
        acc += half.grads
    acc *= 0.5
    rel = float((acc - g_full).norm() / g_full.norm())
    assert rel < 2e-5, rel  # fp32 summation order only (measured 4e-7 ... 1.3e-6 on the BASELINE configs)
    # and the backward pass repeats: weight gradients bitwise (fixed-order sums), the rest to fp32 summation order
    eng.forward(img)
    eng.loss_backward(lab, label_smoothing=0.1)
    torch.cuda.synchronize()
    _assert_backward_repeats(eng, eng.grads, g_full)


def test_loss_descends_on_a_fixed_batch(deit_b):
    cfg, eng, img, lab = deit_b
    eng.init_params(42)
    eng.adam_m = eng.adam_v = None
    eng.step_count = 0
    losses = []
    for _ in range(8):
        eng.forward(img)
        losses.append(float(eng.loss_backward(lab, label_smoothing=0.1)))
        eng.optimizer_step(lr=1e-3, weight_decay=1e-4, max_norm=1.0)
    assert abs(losses[0] - math.log(1000.0)) < 1e-3
    assert all(math.isfinite(v) for v in losses)
    assert losses[-1] < losses[0] - 0.5, losses


def test_vit_large_384_full_width_small_batch():
    """BASELINE config 5 dimensions (d 1024, 16 heads, N 577, 24 layers) at a batch the test can afford."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import savit_amd  # noqa: F401
    from savit_amd.config import get_config
    from savit_amd.engine import ViTEngine

    cfg = get_config("vit_l_patch16", img_size=384)
    assert cfg.seq_len == 577
    eng = ViTEngine(cfg, 4)
    eng.init_params(1)
    g = torch.Generator(device="cuda").manual_seed(2)
    img = torch.randn(4, 384, 384, 3, device="cuda", generator=g).to(torch.bfloat16)
    lab = torch.randint(0, 1000, (4,), device="cuda", generator=g, dtype=torch.int32)
    assert float(eng.forward(img).abs().max()) == 0.0  # zero head
    assert abs(float(eng.loss_backward(lab)) - math.log(1000.0)) < 1e-4
    _head(eng, cfg)
    full = eng.forward(img).clone()
    assert torch.isfinite(full).all()
    assert torch.equal(eng.forward(img.flip(0).contiguous()), full.flip(0))
    eng.loss_backward(lab)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.grads).all() and float(eng.grads.abs().max()) > 0


def test_cait_s24_full_size_properties():
    """BASELINE config 4 (CaiT-S24, 256 images): zero-head known answers and per-sample independence in eval mode (cait.py:140-183;
    stochastic depth off: stochastic_depth.py:13-14)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import savit_amd  # noqa: F401
    from savit_amd.cait_engine import CaiTEngine
    from savit_amd.config import get_config

    cfg = get_config("cait_s_24")
    B = 256
    eng = CaiTEngine(cfg, B)
    eng.init_params(3)
    g = torch.Generator(device="cuda").manual_seed(4)
    img = torch.randn(B, 224, 224, 3, device="cuda", generator=g).to(torch.bfloat16)
    lab = torch.randint(0, 1000, (B,), device="cuda", generator=g, dtype=torch.int32)
    assert float(eng.forward(img, is_training=False).abs().max()) == 0.0
    assert abs(float(eng.loss_backward(lab)) - math.log(1000.0)) < 1e-4
    # away from the reference's init (zero cls token, LayerScale 1e-6, zero head: the logits would stay ~1e-4): every parameter
    # perturbed, so LayerScale / talking-heads / class-attention all carry signal
    gp = torch.Generator(device="cuda").manual_seed(5)
    eng.params.add_(0.05 * torch.randn(eng.params.shape, device="cuda", generator=gp))
    eng.weights_stale = True
    full = eng.forward(img, is_training=False).clone()
    assert torch.isfinite(full).all() and float(full.abs().max()) > 0.05
    assert torch.equal(eng.forward(img, is_training=False), full)
    perm = torch.randperm(B, device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))
    assert torch.equal(eng.forward(img[perm].contiguous(), is_training=False), full[perm])
    eng.forward(img, is_training=True)
    eng.loss_backward(lab)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.grads).all() and float(eng.grads.abs().max()) > 0
    # a 16-image engine reproduces rows of the 256-image batch, and the mean gradient is the mean of the two half-batch gradients
    # (eval mode: stochastic depth is the identity, so the identity is exact up to fp32 summation order) - as for the ViT configs
    small = CaiTEngine(cfg, 16)
    small.params, small.w, small.weights_stale = eng.params, eng.w, False
    got = small.forward(img[64:80].contiguous(), is_training=False)
    err = float((got - full[64:80]).norm() / full[64:80].norm())
    assert err < 1e-6, err
    del small
    eng.forward(img, is_training=False)
    eng.loss_backward(lab)
    torch.cuda.synchronize()
    g_full = eng.grads.clone()
    eng.forward(img, is_training=False)
    eng.loss_backward(lab)
    torch.cuda.synchronize()
    _assert_backward_repeats(eng, eng.grads, g_full)
    half = CaiTEngine(cfg, B // 2)
    half.params, half.w, half.weights_stale = eng.params, eng.w, False
    acc = torch.zeros_like(g_full)
    for lo in (0, B // 2):
        half.forward(img[lo:lo + B // 2].contiguous(), is_training=False)
        half.loss_backward(lab[lo:lo + B // 2].contiguous())
        torch.cuda.synchronize()
        acc += half.grads
    acc *= 0.5
    rel = float((acc - g_full).norm() / g_full.norm())
    print(f"[cait_s_24 B={B}] sub-batch rows {err:.2e}, half-batch gradient linearity {rel:.2e}")
    assert rel < 2e-5, rel


def _full_workload_properties(model, img_size, B, sub, seed):
    """Size-independent properties at a BASELINE config's FULL per-GPU workload: zero-head known answers, bitwise repeatability,
    permutation equivariance, a `sub`-image engine reproducing rows of the full batch, finite non-zero gradients, and the mean
    gradient equal to the mean of the two half-batch gradients."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import savit_amd  # noqa: F401
    from savit_amd.config import get_config
    from savit_amd.engine import ViTEngine

    cfg = get_config(model, img_size=img_size)
    eng = ViTEngine(cfg, B)
    eng.init_params(seed)
    g = torch.Generator(device="cuda").manual_seed(seed + 1)
    img = torch.randn(B, img_size, img_size, 3, device="cuda", generator=g).to(torch.bfloat16)
    lab = torch.randint(0, 1000, (B,), device="cuda", generator=g, dtype=torch.int32)
    assert float(eng.forward(img).abs().max()) == 0.0  # vit.py:96-98
    assert abs(float(eng.loss_backward(lab, label_smoothing=0.1)) - math.log(1000.0)) < 1e-4
    gt = eng.layout.flax_tree(eng.grads)["params"]
    assert float(gt["Dense_0"]["bias"].abs().max()) > 0 and float(gt["PatchEmbedBlock_0"]["Dense_0"]["kernel"].abs().max()) == 0.0
    _head(eng, cfg)
    full = eng.forward(img).clone()
    assert torch.isfinite(full).all() and float(full.abs().max()) > 0.1
    assert torch.equal(eng.forward(img), full)
    perm = torch.randperm(B, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed + 2))
    assert torch.equal(eng.forward(img[perm].contiguous()), full[perm])
    eng.forward(img)
    eng.loss_backward(lab, label_smoothing=0.1)
    torch.cuda.synchronize()
    g_full = eng.grads.clone()
    assert torch.isfinite(g_full).all() and float(g_full.abs().max()) > 0
    small = ViTEngine(cfg, sub)
    small.params, small.w, small.weights_stale = eng.params, eng.w, False
    lo = B // 4
    got = small.forward(img[lo:lo + sub].contiguous())
    err = float((got - full[lo:lo + sub]).norm() / full[lo:lo + sub].norm())
    assert err < 1e-6, err  # measured 0: same K order whatever the grid
    del small
    half = ViTEngine(cfg, B // 2)
    half.params, half.w, half.weights_stale = eng.params, eng.w, False
    acc = torch.zeros_like(g_full)
    for lo in (0, B // 2):
        half.forward(img[lo:lo + B // 2].contiguous())
        half.loss_backward(lab[lo:lo + B // 2].contiguous(), label_smoothing=0.1)
        torch.cuda.synchronize()
        acc += half.grads
    acc *= 0.5
    rel = float((acc - g_full).norm() / g_full.norm())
    print(f"[{model} {img_size} B={B}] sub-batch rows {err:.2e}, half-batch gradient linearity {rel:.2e}")
    assert rel < 2e-5, rel  # measured 3.8e-7 / 1.3e-6: fp32 summation order only


def test_deit_small_full_workload():
    """BASELINE config 2: DeiT-S/16 224^2 at 256 images."""
    _full_workload_properties("vit_s_patch16", 224, 256, 16, 11)


def test_vit_large_384_full_workload():
    """BASELINE config 5: ViT-L/16 at 384^2, 256 images per GPU (N = 577, 24 layers; ~130 GB of saved activations)."""
    _full_workload_properties("vit_l_patch16", 384, 256, 8, 21)
