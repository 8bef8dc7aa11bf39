"""MLP-Mixer through the HIP engine (/root/reference/models/mlp_mixer.py) vs the oracle (GPU).

Kernel-level: the per-image transpose (with its fused residual add and row sums) and the token mean are bit-exact or
one-rounding-exact against numpy.  Model-level: same bars as tests/test_model_gpu.py - bf16 logits vs the fp32 oracle
no worse than ~2.5x the bf16-emulating oracle's own deviation; every parameter gradient vs fp32 autograd."""
import math

import numpy as np
import pytest
import torch

from oracle import torch_ref, vit_ref
from tests import parity_bars

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import savit_amd
    from savit_amd import config, mixer_engine, model, ops  # noqa: F401

    return savit_amd


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _flat(tree):
    return {k: v.detach().float().cpu().numpy() for k, v in torch_ref.leaves(tree)}


def _bf(x):
    return torch.as_tensor(vit_ref.bf16_round(np.asarray(x, np.float32))).cuda().to(torch.bfloat16)


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("B,R,Cc,ld_src,ld_dst", [(3, 196, 128, 128, 256), (2, 49, 64, 64, 64), (2, 128, 196, 256, 128), (1, 70, 72, 80, 72),
                                                 (2, 16, 128, 128, 64)])
def test_transpose_bf16(pkg, B, R, Cc, ld_src, ld_dst):
    from savit_amd import ops

    rng = np.random.default_rng(B * 1000 + R)
    src = np.zeros((B, R, ld_src), np.float32)
    src[:, :, :Cc] = vit_ref.bf16_round(rng.standard_normal((B, R, Cc)).astype(np.float32))
    s = _bf(src)
    dst = torch.full((B, Cc, ld_dst), 7.0, device="cuda", dtype=torch.bfloat16)
    rowsum = torch.full((R,), 0.5, device="cuda") if R % 4 == 0 else None  # the wrapper reduces a [tiles, R] slab of partial sums
    ops.transpose_bf16(s, dst, R=R, Cc=Cc, rowsum=rowsum)
    got = dst.float().cpu().numpy()
    assert np.array_equal(got[:, :, :R], np.swapaxes(src[:, :, :Cc], 1, 2))
    assert np.all(got[:, :, R:] == 7.0)  # columns beyond R are not written
    if rowsum is not None:  # accumulated onto what the buffer held
        want = 0.5 + src[:, :, :Cc].astype(np.float64).sum(axis=(0, 2))
        assert np.allclose(rowsum.cpu().numpy(), want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("round_out", [False, True])
def test_transpose_add_residual(pkg, round_out):
    from savit_amd import ops

    B, R, Cc, ld_src = 3, 128, 196, 256  # src = token-mixing branch [B, d, l(+pad)], out = residual stream [B, l, d]
    rng = np.random.default_rng(4)
    src = vit_ref.bf16_round(rng.standard_normal((B, R, ld_src)).astype(np.float32))
    resid = rng.standard_normal((B, Cc, R)).astype(np.float32)
    out = torch.empty(B, Cc, R, device="cuda")
    ops.transpose_bf16(_bf(src), None, R=R, Cc=Cc, resid=torch.as_tensor(resid).cuda(), out_f32=out, round_out_bf16=round_out)
    want = resid + np.swapaxes(src[:, :, :Cc], 1, 2)
    if round_out:
        want = vit_ref.bf16_round(want)
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("B,L,d", [(3, 196, 512), (2, 49, 768), (2, 16, 128), (1, 5, 64)])
def test_token_mean(pkg, B, L, d):
    from savit_amd import ops

    rng = np.random.default_rng(L)
    h = vit_ref.bf16_round(rng.standard_normal((B, L, d)).astype(np.float32))
    z = ops.token_mean_fwd(_bf(h)).float().cpu().numpy()
    want = h.astype(np.float64).mean(axis=1)
    assert np.abs(z - want).max() <= 2.0 ** -8 * np.abs(want).max() + 1e-6  # one bf16 rounding of an fp32 mean
    dz = vit_ref.bf16_round(rng.standard_normal((B, d)).astype(np.float32))
    dh = ops.token_mean_bwd(_bf(dz), L).float().cpu().numpy()
    assert dh.shape == (B, L, d)
    assert np.array_equal(dh, np.broadcast_to(vit_ref.bf16_round(dz * np.float32(1.0 / L))[:, None, :], dh.shape))


# ------------------------------------------------------------------------------------------------ model
CASES = {
    # 16 tokens (hidden 8): both token widths far below the 64-element padding
    "tiny": dict(kind="mixer", num_layers=2, num_heads=1, embed_dim=128, patch=8, num_classes=16, img_size=32),
    # the /16 geometry of create_model: 196 tokens, hidden 98
    "p16": dict(kind="mixer", num_layers=2, num_heads=1, embed_dim=128, patch=16, num_classes=104, img_size=224),
    # the /32 geometry: 49 tokens, hidden 24
    "p32": dict(kind="mixer", num_layers=1, num_heads=1, embed_dim=192, patch=32, num_classes=1000, img_size=224),
}


def _cfgs(**kw):
    from savit_amd.config import ModelConfig

    return ModelConfig(**kw), vit_ref.Cfg(**kw)


@pytest.mark.parametrize("case,B", [("tiny", 3), ("p16", 2), ("p32", 4)])
def test_forward_backward_parity(pkg, case, B):
    from savit_amd.mixer_engine import MixerEngine

    mc, oc = _cfgs(**CASES[case])
    rng = np.random.default_rng(21)
    params = vit_ref.init_params(oc, seed=6, randomize=True)
    images = vit_ref.bf16_round(rng.standard_normal((B, oc.img_size, oc.img_size, 3)).astype(np.float32))
    labels = rng.integers(0, oc.num_classes, B)
    eng = MixerEngine(mc, B)
    eng.load_params(params)
    logits = eng.forward(torch.as_tensor(images).cuda()).float().cpu().numpy()
    ref32 = vit_ref.forward(params, images, oc, mode="f32")
    refbf = vit_ref.forward(params, images, oc, mode="bf16")
    parity_bars.check_logits(f"mixer:{case}", logits, ref32, refbf)
    loss = float(eng.loss_backward(torch.as_tensor(labels).cuda(), 0.1))
    loss_ref, _, grads_ref = torch_ref.loss_and_grads(params, images, labels, oc, 0.1)
    assert abs(loss - loss_ref) < 2e-2 * max(1.0, abs(loss_ref))
    got = _flat(eng.grad_tree()["params"])
    assert set(got) == set(grads_ref)
    zero_grad = []
    for k, g in grads_ref.items():
        assert got[k].shape == g.shape, k
        if k.endswith("FFBlock_0/Dense_1/bias"):
            # known answer: this bias shifts every channel of a token by the same amount and every later LayerNorm removes such
            # a shift, so its exact gradient is 0 (tests/test_mixer_oracle.py); both sides hold rounding noise only
            scale = np.linalg.norm(grads_ref[k.replace("Dense_1/bias", "Dense_0/bias")])
            assert np.linalg.norm(g) < 1e-4 * scale and np.linalg.norm(got[k]) < 2e-2 * scale, (k, np.linalg.norm(got[k]), scale)
            zero_grad.append(k)
    parity_bars.check_grads(f"mixer:{case}", got, grads_ref, skip=zero_grad)
    assert abs(loss - vit_ref.loss_fn(logits, labels, 0.1)) < 1e-4 * max(1.0, abs(loss))


def test_padding_is_invisible(pkg):
    """The token-mixing parameters are stored padded to multiples of 64: the padding must hold zeros, receive exactly zero
    gradient, and stay zero through AdamW steps with weight decay."""
    from savit_amd.mixer_engine import MixerEngine

    mc, oc = _cfgs(**CASES["p16"])
    B = 2
    eng = MixerEngine(mc, B)
    eng.init_params(3)
    lay = eng.layout
    assert (lay.Lp, lay.Fp) == (256, 128) and oc.tokens_hidden == 98
    x = torch.randn(B, 224, 224, 3, device="cuda")
    y = torch.tensor([1, 5], device="cuda")

    def pad_abs_max(flat):
        worst = 0.0
        for l in range(mc.num_layers):
            for nme in ("tW1", "tb1", "tW2", "tb2"):
                o, shape = lay.off[f"l{l}.{nme}"]
                k = int(np.prod(shape))
                full = flat[o:o + k].view(*shape).clone()
                lv = full
                for ax, s in enumerate(lay.logical[f"l{l}.{nme}"]):
                    lv = lv.narrow(ax, 0, s)
                lv.zero_()
                worst = max(worst, float(full.abs().max()))
        return worst

    assert pad_abs_max(eng.params) == 0.0
    for _ in range(3):
        eng.forward(x)
        eng.loss_backward(y, 0.1)
        assert pad_abs_max(eng.grads) == 0.0
        eng.optimizer_step(lr=1e-2, weight_decay=0.05, max_norm=1.0)
    assert pad_abs_max(eng.params) == 0.0
    assert float(lay.view(eng.grads, "l0.tW1").abs().max()) > 0 and float(lay.view(eng.grads, "l1.tb2").abs().max()) > 0


def test_create_model_protocol_and_tree(pkg):
    """create_model names (create_model.py:184-213), Flax-shaped tree, logits shape, init loss."""
    from savit_amd.model import create_model

    model = create_model("mixer_s_patch32", dtype=torch.bfloat16)
    x = torch.randn(2, 224, 224, 3, device="cuda")
    logits, params = model.init_with_output(0, x, is_training=True)
    assert tuple(logits.shape) == (2, 1000) and logits.dtype == torch.bfloat16
    p = params["params"]
    blk = p["MixerBlock_7"]
    assert tuple(blk["FFBlock_0"]["Dense_0"]["kernel"].shape) == (49, 24)
    assert tuple(blk["FFBlock_0"]["Dense_1"]["kernel"].shape) == (24, 49)
    assert tuple(blk["FFBlock_0"]["Dense_1"]["bias"].shape) == (49,)
    assert tuple(blk["FFBlock_1"]["Dense_0"]["kernel"].shape) == (512, 2048)
    assert tuple(p["PatchEmbedBlock_0"]["Dense_0"]["bias"].shape) == (512,)
    shapes = vit_ref.param_shapes(vit_ref.get_cfg("mixer_s_patch32"))
    assert {"params/" + k: tuple(v.shape) for k, v in torch_ref.leaves(p)} == shapes
    assert sum(v.numel() for _, v in torch_ref.leaves(p)) == 18_920_880
    # apply() with a caller-owned tree gives the same logits; is_training changes nothing
    again = model.apply(params, x, is_training=False)
    assert torch.equal(again, logits)
    for name in ("mixer_s_patch16", "mixer_b_patch32", "mixer_b_patch16", "mixer_l_patch32", "mixer_l_patch16"):
        assert create_model(name, dtype=torch.bfloat16).cfg.kind == "mixer"


def test_overlapped_backward_equals_serial(pkg):
    """Side-stream weight-gradient launches must give the serial plan's gradients (fp32 atomics reorder: tight tolerance)."""
    from savit_amd.mixer_engine import MixerEngine

    mc, _ = _cfgs(**CASES["p16"])
    B = 4
    eng = MixerEngine(mc, B)
    eng.init_params(1)
    x = torch.randn(B, 224, 224, 3, device="cuda")
    y = torch.randint(0, mc.num_classes, (B,), device="cuda")
    eng.overlap_wgrad = True
    eng.forward(x)
    eng.loss_backward(y, 0.1)
    g1 = eng.grads.clone()
    eng.overlap_wgrad = False
    eng.forward(x)
    eng.loss_backward(y, 0.1)
    torch.cuda.synchronize()
    denom = float(g1.abs().max())
    assert float((eng.grads - g1).abs().max()) <= 1e-5 * max(denom, 1.0)


def test_flax_checkpoint_round_trip(pkg, tmp_path):
    from savit_amd import flax_ckpt
    from savit_amd.mixer_engine import MixerEngine

    mc, _ = _cfgs(**CASES["tiny"])
    a = MixerEngine(mc, 2)
    a.init_params(9)
    x = torch.randn(2, 32, 32, 3, device="cuda")
    y = torch.tensor([1, 2], device="cuda")
    a.forward(x)
    a.loss_backward(y, 0.1)
    a.optimizer_step(lr=1e-3, weight_decay=0.01)
    path = flax_ckpt.save_from_engine(a, str(tmp_path), 1)
    b = MixerEngine(mc, 2)
    assert flax_ckpt.load_into_engine(b, flax_ckpt.read_train_state(path)) == 1
    for (k, va), (_, vb) in zip(torch_ref.leaves(a.param_tree()), torch_ref.leaves(b.param_tree())):
        assert torch.equal(va, vb), k
    assert torch.equal(a.forward(x), b.forward(x))


def test_tall_colsum_finalize(pkg):
    """The token GEMMs produce tall, narrow partial-sum slabs (>= 1024 rows x 64..256 columns): reduced in row chunks."""
    from savit_amd import ops

    rng = np.random.default_rng(3)
    slab = rng.standard_normal((1536, 256)).astype(np.float32)
    out = torch.full((256,), 1.5, device="cuda")
    ops.colsum_finalize(torch.as_tensor(slab).cuda(), out, accumulate=True)
    want = 1.5 + slab.astype(np.float64).sum(0)
    assert np.allclose(out.cpu().numpy(), want, rtol=1e-5, atol=1e-4)
    out2 = torch.full((256,), 9.0, device="cuda")
    ops.colsum_finalize(torch.as_tensor(slab).cuda(), out2, accumulate=False)  # overwrite mode keeps the single-chunk path
    assert np.allclose(out2.cpu().numpy(), want - 1.5, rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------------------------------------ full size (Mixer-B/16, 128 images)
@pytest.fixture(scope="module")
def mixer_b(pkg):
    from savit_amd.config import get_config
    from savit_amd.mixer_engine import MixerEngine

    cfg = get_config("mixer_b_patch16")
    eng = MixerEngine(cfg, 128)
    eng.init_params(42)
    g = torch.Generator(device="cuda").manual_seed(1)
    img = torch.randn(128, 224, 224, 3, device="cuda", generator=g).to(torch.bfloat16)
    lab = torch.randint(0, 1000, (128,), device="cuda", generator=g, dtype=torch.int32)
    return cfg, eng, img, lab


def test_full_size_rows_independent_and_equivariant(mixer_b):
    """No BatchNorm on this path either: token mixing runs inside one image.  Size-independent properties at 128 images."""
    from savit_amd.mixer_engine import MixerEngine

    cfg, eng, img, lab = mixer_b
    full = eng.forward(img).clone()
    assert torch.isfinite(full).all() and float(full.abs().max()) > 0.1
    assert torch.equal(eng.forward(img), full)  # repeatable bit for bit
    perm = torch.randperm(128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    assert torch.equal(eng.forward(img[perm].contiguous()), full[perm])
    small = MixerEngine(cfg, 16)
    small.params, small.w, small.weights_stale = eng.params, eng.w, False
    sub = small.forward(img[32:48].contiguous())
    err = float((sub - full[32:48]).norm() / full[32:48].norm())
    assert err < 2e-3, err


def test_full_size_gradient_linearity_and_descent(mixer_b):
    """grad(mean over 128) == mean of the two 64-image gradients (train.py:96), and a fixed batch's loss goes down."""
    from savit_amd.mixer_engine import MixerEngine

    cfg, eng, img, lab = mixer_b
    eng.forward(img)
    l0 = float(eng.loss_backward(lab, label_smoothing=0.1))
    g_full = eng.grads.clone()
    half = MixerEngine(cfg, 64)
    half.params, half.w, half.weights_stale = eng.params, eng.w, False
    acc = torch.zeros_like(g_full)
    for k in range(2):
        half.forward(img[64 * k:64 * (k + 1)].contiguous())
        half.loss_backward(lab[64 * k:64 * (k + 1)].contiguous(), label_smoothing=0.1)
        acc += half.grads
    acc *= 0.5
    err = float((acc - g_full).norm() / g_full.norm())
    assert err < 2e-2, err
    assert abs(l0 - 7.0) < 1.0  # lecun-normal head on unit-variance features: close to ln(1000) = 6.9
    for _ in range(6):
        eng.optimizer_step(lr=1e-3, weight_decay=1e-4, max_norm=1.0)
        eng.forward(img)
        l1 = float(eng.loss_backward(lab, label_smoothing=0.1))
    assert math.isfinite(l1) and l1 < l0 - 0.05, (l0, l1)


def test_transpose_bf16_jobs_is_the_single_launches(pkg):
    """savit_transpose_bf16_jobs (round 5: the ViT engines' operand refresh in one launch) = the same transposes launched one by one,
    bit for bit: batched jobs with layer strides, a ragged one (rows, cols not multiples of 64), a job with batch 0, and the limits."""
    import ctypes

    from savit_amd import lib as _lib

    L = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(3, 768, 2304, 2304 + 64, 768), (3, 768, 768, 768, 776), (2, 100, 72, 72, 104), (0, 64, 64, 64, 64), (1, 768, 384, 384, 768)]
    jobs = (_lib.TransposeJob * len(shapes))()
    keep, want = [], []
    s0 = torch.cuda.current_stream().cuda_stream
    for q, (B, R, Cc, ld_src, ld_dst) in zip(jobs, shapes):
        nb = max(B, 1)
        src = torch.randn(nb, R, ld_src, device="cuda", generator=g).to(torch.bfloat16)
        dst = torch.full((nb, Cc, ld_dst), 3.0, dtype=torch.bfloat16, device="cuda")
        ref = torch.full((nb, Cc, ld_dst), 3.0, dtype=torch.bfloat16, device="cuda")
        if B:
            assert L.savit_transpose_bf16(src.data_ptr(), R * ld_src, ld_src, ref.data_ptr(), Cc * ld_dst, ld_dst, B, R, Cc, None, None, 0, None, 0, s0) == 0
            assert torch.equal(ref[:, :, :R], src[:, :, :Cc].transpose(1, 2)) and bool((ref[:, :, R:] == 3.0).all())
        q.src, q.dst, q.src_batch_stride, q.dst_batch_stride = src.data_ptr(), dst.data_ptr(), R * ld_src, Cc * ld_dst
        q.ld_src, q.ld_dst, q.batch, q.rows, q.cols = ld_src, ld_dst, B, R, Cc
        keep.append((src, dst))
        want.append(ref)
    assert L.savit_transpose_bf16_jobs(jobs, len(shapes), s0) == 0
    torch.cuda.synchronize()
    for (src, dst), ref in zip(keep, want):
        assert torch.equal(dst, ref)
    assert L.savit_transpose_bf16_jobs(jobs, 9, s0) == _lib.SAVIT_EINVAL
    assert L.savit_transpose_bf16_jobs(None, 0, s0) == 0
    jobs[2].ld_dst = 96  # < rows
    assert L.savit_transpose_bf16_jobs(jobs, len(shapes), s0) == _lib.SAVIT_EINVAL
