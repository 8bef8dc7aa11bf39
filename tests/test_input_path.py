"""Input path: normalise / mixup / cutmix kernels against the numpy restatement of the reference's TF ops
(oracle/vit_ref.py), plus known answers of the formulas.  GPU tests call through the C ABI (savit_amd.ops)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vit_ref  # noqa: E402


def rb(x):  # round to bf16, as float32
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


# ------------------------------------------------------------------------------------------------ CPU: the formulas
def test_mixup_weight_known_answers():
    # augment_ops.py:169-172 with beta = 0.8: u=1 -> 0.5 -> max(0.5, 0.5); u=0 -> 0 -> 1; always in [0.5, 1]
    assert vit_ref.mixup_weight([1.0], 0.8)[0] == 0.5
    assert vit_ref.mixup_weight([0.0], 0.8)[0] == 1.0
    w = vit_ref.mixup_weight(np.linspace(0, 1, 101), 0.8)
    assert (w >= 0.5).all() and (w <= 1.0).all()
    np.testing.assert_allclose(vit_ref.mixup_weight([0.5], 1.0), [0.75])


def test_cutmix_box_known_answers():
    # u = 0.5, beta = 1: weight 0.25, ratio 0.5, box 112 x 112; shifts are taken modulo (224 - 112)
    w, box = vit_ref.cutmix_box([0.5], [100 + 112], [7], 224, 224)
    assert w[0] == 0.25 and box.tolist() == [[7, 119, 100, 212]]
    # the box never leaves the image and its area fraction is <= the label weight (int truncation)
    rng = np.random.default_rng(0)
    u = rng.random(256)
    w, box = vit_ref.cutmix_box(u, rng.integers(0, 224, 256), rng.integers(0, 224, 256), 224, 224)
    assert (box[:, 0] >= 0).all() and (box[:, 1] <= 224).all() and (box[:, 2] >= 0).all() and (box[:, 3] <= 224).all()
    area = (box[:, 1] - box[:, 0]) * (box[:, 3] - box[:, 2]) / (224 * 224)
    assert (area <= w + 1e-12).all() and (w <= 0.5).all()


def test_cutmix_apply_matches_where_semantics():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((4, 8, 8, 3)).astype(np.float32)
    oh = np.eye(5, dtype=np.float32)[[0, 1, 2, 3]]
    w, box = vit_ref.cutmix_box([0.5, 0.18, 0.98, 0.02], [1, 2, 3, 4], [4, 3, 2, 1], 8, 8)
    xo, lo = vit_ref.batch_cutmix_apply(x, oh, w, box)
    for b in range(4):
        y0, y1, x0, x1 = box[b]
        ref = x[3 - b].copy()
        ref[y0:y1, x0:x1] = x[b, y0:y1, x0:x1]  # inside the box the sample keeps its OWN pixels (augment_ops.py:136-138)
        np.testing.assert_array_equal(xo[b], ref)
    np.testing.assert_allclose(lo.sum(1), 1.0, rtol=1e-6)
    np.testing.assert_allclose(lo[0, 0], w[0], rtol=1e-6)


def test_label_mix_equals_two_label_loss():
    """Mixing one-hot rows (the reference's pipeline) == ratio*CE(y) + (1-ratio)*CE(y1) (train.py:83-88), which is what the
    GPU path hands to the loss kernel."""
    rng = np.random.default_rng(2)
    B, C = 6, 10
    logits = rng.standard_normal((B, C))
    labels = rng.integers(0, C, B)
    index = rng.permutation(B)
    mix = vit_ref.mixup_weight(rng.random(B), 0.8).astype(np.float32)
    oh = np.eye(C, dtype=np.float32)[labels]
    _, lm = vit_ref.batch_mixup_apply(np.zeros((B, 1, 1, 8), np.float32), oh, mix, index)
    logp = logits - np.log(np.exp(logits).sum(1, keepdims=True))
    a = -(lm * logp).sum(1)
    b = -(mix * logp[np.arange(B), labels] + (1 - mix) * logp[np.arange(B), labels[index]])
    np.testing.assert_allclose(a, b, rtol=1e-6)


# ------------------------------------------------------------------------------------------------ GPU: the kernels
@pytest.fixture(scope="module")
def ops():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import savit_amd  # noqa: F401
    from savit_amd import ops as _ops
    return _ops


@pytest.mark.gpu
@pytest.mark.parametrize("layout,dtype", [("NHWC", "f32"), ("NHWC", "u8"), ("HWCN", "f32")])
@pytest.mark.parametrize("N,S", [(3, 32), (5, 224), (70, 16)])
def test_normalize_bit_exact(ops, layout, dtype, N, S):
    import torch
    rng = np.random.default_rng(N * S)
    if dtype == "u8":
        img = rng.integers(0, 256, (N, S, S, 3), dtype=np.uint8)
        scale = 1.0 / 255.0
    else:
        img = rng.random((N, S, S, 3), dtype=np.float32)
        scale = 1.0
    ref = rb(vit_ref.normalize_images(img, ops.IMAGENET_1K_MEAN, ops.IMAGENET_1K_STD, scale))
    src = np.ascontiguousarray(np.transpose(img, (1, 2, 3, 0))) if layout == "HWCN" else img
    out = ops.normalize_to_nhwc_bf16(torch.from_numpy(src).cuda(), layout=layout)
    got = out.float().cpu().numpy()
    # the kernel multiplies by 1/std (one fp32 rounding away from the oracle's form at most): allow one bf16 ulp
    assert got.shape == (N, S, S, 3)
    assert np.abs(got - ref).max() <= np.abs(ref).max() * 2.0 ** -7
    assert (got == ref).mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("B,S", [(2, 32), (8, 224), (5, 48)])
def test_mixup_bit_exact(ops, B, S):
    import torch
    rng = np.random.default_rng(B + S)
    x = rb(rng.standard_normal((B, S, S, 3)))
    mix = vit_ref.mixup_weight(rng.random(B), 0.8).astype(np.float32)
    index = rng.permutation(B).astype(np.int32)
    ref, _ = vit_ref.batch_mixup_apply(x, np.zeros((B, 1), np.float32), mix, index)
    out = ops.batch_mixup(torch.from_numpy(x).cuda().to(torch.bfloat16), torch.from_numpy(mix).cuda(), torch.from_numpy(index).cuda())
    got = out.float().cpu().numpy()
    # fp32 a*w + c*(1-w) may be contracted into an fma on the GPU: equal after bf16 rounding except at rounding ties
    assert np.abs(got - rb(ref)).max() <= np.abs(ref).max() * 2.0 ** -7
    assert (got == rb(ref)).mean() > 0.995


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(2, 16, 16), (8, 224, 224), (5, 40, 24)])
def test_cutmix_bit_exact(ops, B, H, W):
    import torch
    rng = np.random.default_rng(B * H)
    x = rb(rng.standard_normal((B, H, W, 3)))
    w, box = vit_ref.cutmix_box(rng.random(B), rng.integers(0, W, B), rng.integers(0, H, B), H, W)
    ref, _ = vit_ref.batch_cutmix_apply(x, np.zeros((B, 1), np.float32), w, box)
    index = np.arange(B - 1, -1, -1, dtype=np.int32)
    out = ops.batch_cutmix(torch.from_numpy(x).cuda().to(torch.bfloat16), torch.from_numpy(box.astype(np.int32)).cuda(), torch.from_numpy(index).cuda())
    np.testing.assert_array_equal(out.float().cpu().numpy(), ref)  # a pure select: exact


@pytest.mark.gpu
def test_mix_batch_sampler_and_contracts(ops):
    import torch
    from savit_amd import augment
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(8, 32, 32, 3, device="cuda").to(torch.bfloat16)
    y = torch.arange(8, device="cuda", dtype=torch.int32)
    seen = set()
    for _ in range(12):
        xo, yo, ym, ratio = augment.mix_batch(x, y, generator=g)
        assert xo.shape == x.shape and ym is not None and ratio.shape == (8,)
        r = ratio.cpu().numpy()
        seen.add("mixup" if (r >= 0.5).all() and (r > 0.5).any() else "cutmix")
        assert ((r >= 0) & (r <= 1)).all()
    assert seen == {"mixup", "cutmix"}  # both branches are drawn (augment_utils.py:118-122)
    xo, yo, ym, ratio = augment.mix_batch(x, y, prob_to_apply=0.0)
    assert ym is None and ratio is None and xo.data_ptr() == x.data_ptr()
    w, box, idx = augment.sample_cutmix(64, 224, 224, generator=g)
    b = box.cpu().numpy()
    assert (b[:, 0] >= 0).all() and (b[:, 1] <= 224).all() and (b[:, 2] >= 0).all() and (b[:, 3] <= 224).all()
    with pytest.raises(ValueError):
        ops.batch_mixup(x, torch.ones(8, device="cuda"), torch.full((8,), 9, device="cuda", dtype=torch.int32))
    with pytest.raises(ValueError):
        ops.batch_mixup(x, torch.ones(8, device="cuda"), torch.zeros(8, device="cuda", dtype=torch.int32), out=x)
    with pytest.raises(ValueError):
        ops.normalize_to_nhwc_bf16(torch.zeros(2, 8, 8, 3))  # CPU tensor: no CPU path
