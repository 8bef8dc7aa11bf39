"""End-to-end parity of the HIP engine behind the reference's create_model()/init/apply API (GPU).

Forward: bf16 logits vs the oracle's fp32 math on identical weights/inputs.  Per SURVEY A.5 the reference's own
bf16 graph deviates O(1e-2) from fp32 after a few layers, so the end-to-end bar is: relative L2 vs fp32 oracle no
worse than ~2x the bf16-emulating oracle's own deviation (printed), per-kernel bars are in test_kernels_gpu.py.
Backward: every parameter gradient vs fp32 autograd of the independent torch composition."""
import math
import os

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
    from savit_amd import config, engine, model  # noqa: F401

    return savit_amd


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _cfgs(pkg, **kw):
    from savit_amd.config import ModelConfig

    return ModelConfig(**kw), vit_ref.Cfg(**kw)


def _flat(tree):
    return {k: v.detach().float().cpu().numpy() for k, v in torch_ref.leaves(tree)}


CASES = {
    "tiny": dict(kind="vit", num_layers=2, num_heads=2, embed_dim=128, patch=8, num_classes=16, img_size=32),
    "ti2": dict(kind="vit", num_layers=2, num_heads=3, embed_dim=192, patch=16, num_classes=1000, img_size=224),
    "s1_p32": dict(kind="vit", num_layers=1, num_heads=6, embed_dim=384, patch=32, num_classes=1000, img_size=224),
    # ViT-L/16-at-384 geometry (N = 577 tokens: the online-softmax attention kernels) at a small width
    "n577": dict(kind="vit", num_layers=1, num_heads=2, embed_dim=128, patch=16, num_classes=16, img_size=384),
    # ViT-B/16-at-512 geometry (N = 1 025 tokens: the STREAMING attention kernels, round 5 - the resident ones stop at 608) at a small width
    "n1025": dict(kind="vit", num_layers=1, num_heads=2, embed_dim=128, patch=16, num_classes=16, img_size=512),
    # head_dim 48 (the CaiT head width) through the ViT engine
    "hd48": dict(kind="vit", num_layers=2, num_heads=4, embed_dim=192, patch=8, num_classes=16, img_size=32),
}


@pytest.mark.parametrize("plan", ["default", "dense"])
@pytest.mark.parametrize("case,B", [("tiny", 3), ("ti2", 4), ("s1_p32", 2), ("n577", 2), ("n1025", 2), ("hd48", 3)])
def test_forward_backward_parity(pkg, case, B, plan):
    """plan "default": the product's plan (last layer on the cls rows); "dense" (cls_only_last=False): every token of every layer through
    the dense kernels - for the one-layer cases (s1_p32, n577: the resident general attention kernels, n1025: the streaming ones) the only
    run in which those kernels are compared with the oracle at model level.  Same bars for both."""
    from savit_amd.engine import ViTEngine

    mc, oc = _cfgs(pkg, **CASES[case])
    rng = np.random.default_rng(11)
    params = vit_ref.init_params(oc, seed=5, randomize=True)
    images = vit_ref.bf16_round(rng.standard_normal((B, oc.img_size, oc.img_size, 3)).astype(np.float32))
    labels = rng.integers(0, oc.num_classes, B)
    eng = ViTEngine(mc, B, **({"cls_only_last": False} if plan == "dense" else {}))
    assert eng.cls_only_last == (plan == "default")
    eng.load_params(params)
    logits = eng.forward(torch.as_tensor(images).cuda()).float().cpu().numpy()
    assert sum(1 for c in eng._fwd_plan.calls if c[0] is eng.L.savit_attention_fwd) == oc.num_layers - (1 if eng.cls_fwd else 0)
    ref32 = vit_ref.forward(params, images, oc, mode="f32")
    refbf = vit_ref.forward(params, images, oc, mode="bf16")
    parity_bars.check_logits(f"vit:{case}", logits, ref32, refbf)
    # loss + backward
    loss = float(eng.loss_backward(torch.as_tensor(labels).cuda(), 0.1))
    loss_ref, _, grads_ref = torch_ref.loss_and_grads(params, images, labels, oc, 0.1)
    assert abs(loss - loss_ref) < 2e-2 * max(1.0, abs(loss_ref))
    got = _flat(eng.grad_tree()["params"])
    parity_bars.check_grads(f"vit:{case}", got, grads_ref)
    # the engine's loss must equal the oracle loss evaluated on the engine's own logits
    assert abs(loss - vit_ref.loss_fn(logits, labels, 0.1)) < 1e-4 * max(1.0, abs(loss))


def test_known_answers_zero_head(pkg):
    """reference initialisers => logits == 0, loss == ln(1000), only the head receives gradient."""
    from savit_amd.model import create_model

    model = create_model("vit_ti_patch16", dtype=torch.bfloat16)
    x = torch.randn(2, 224, 224, 3, device="cuda")
    logits, params = model.init_with_output(0, x, is_training=True)
    assert tuple(logits.shape) == (2, 1000) and logits.dtype == torch.bfloat16
    assert float(logits.float().abs().max()) == 0.0
    eng = model.engine(2)
    loss = float(eng.loss_backward(torch.tensor([3, 7], device="cuda"), 0.1))
    assert abs(loss - math.log(1000.0)) < 1e-5
    g = _flat(eng.grad_tree()["params"])
    for k, v in g.items():
        if k in ("Dense_0/bias", "Dense_0/kernel"):
            continue
        assert np.all(v == 0), k
    assert abs(g["Dense_0/bias"][3] - (1e-3 - 0.9001) / 2) < 2e-3  # dlogits is a bf16 cotangent
    assert np.abs(g["Dense_0/kernel"]).max() > 0
    # Flax-shaped tree
    p = params["params"]
    assert tuple(p["Encoder_0"]["EncoderBlock_0"]["SelfAttentionBlock_0"]["queries"]["kernel"].shape) == (192, 3, 64)
    assert tuple(p["Encoder_0"]["EncoderBlock_0"]["SelfAttentionBlock_0"]["DenseGeneral_0"]["kernel"].shape) == (3, 64, 192)
    assert tuple(p["Encoder_0"]["AddAbsPosEmbed_0"]["pos_embed"].shape) == (1, 197, 192)
    n = sum(v.numel() for _, v in torch_ref.leaves(p))
    assert n == 5_708_008
    assert float(p["cls"].abs().max()) == 0 and float(p["Dense_0"]["kernel"].abs().max()) == 0


@pytest.mark.parametrize("name,N", [("vit_b_patch32", 50), ("vit_b_patch16", 197)])
def test_reference_shape_tests(pkg, name, N):
    """models/vit_test.py:13-26: logits (2, 1000) on ones(2,224,224,3)."""
    from savit_amd.model import create_model

    model = create_model(name, dtype=torch.bfloat16)
    assert model.cfg.seq_len == N
    logits, _ = model.init_with_output(0, torch.ones(2, 224, 224, 3, device="cuda"), is_training=True)
    assert tuple(logits.shape) == (2, 1000)
    with pytest.raises(RuntimeError, match="Model not found."):
        create_model("resnet50")


def test_apply_is_functional_and_is_training_invariant(pkg):
    from savit_amd.model import create_model

    mc, oc = _cfgs(pkg, **CASES["tiny"])
    from savit_amd.model import ViT

    model = ViT(mc, dtype=torch.bfloat16)
    pa = vit_ref.init_params(oc, seed=1, randomize=True)
    pb = vit_ref.init_params(oc, seed=2, randomize=True)
    x = torch.randn(3, 32, 32, 3, device="cuda")
    la = model.apply(pa, x, is_training=True).float()
    lb = model.apply(pb, x, is_training=False).float()
    la2 = model.apply(pa, x, is_training=False).float()
    assert torch.equal(la, la2) and not torch.equal(la, lb)
    # the loader's [H, W, C, N] fp32 batch layout (train.py:80-81) gives the same result
    e = model.engine(3)
    l3 = e.forward(x.permute(1, 2, 3, 0).contiguous()).to(torch.bfloat16).float()
    assert torch.equal(l3, la2)


def test_train_steps_track_oracle(pkg):
    """Three AdamW steps (clip 1.0, lr 3e-3: simple_train.py:25-27) on a fixed batch vs the fp32 oracle loop."""
    from savit_amd.engine import ViTEngine

    mc, oc = _cfgs(pkg, **CASES["tiny"])
    rng = np.random.default_rng(3)
    params = vit_ref.init_params(oc, seed=9, randomize=True)
    B = 8
    images = vit_ref.bf16_round(rng.standard_normal((B, 32, 32, 3)).astype(np.float32))
    labels = rng.integers(0, oc.num_classes, B)
    eng = ViTEngine(mc, B)
    eng.load_params(params)
    flat = {k: v.astype(np.float64) for k, v in vit_ref.flatten(params).items()}
    m = {k: np.zeros_like(v) for k, v in flat.items()}
    v = {k: np.zeros_like(x) for k, x in flat.items()}
    img_t, lab_t = torch.as_tensor(images).cuda(), torch.as_tensor(labels).cuda()
    losses, ref_losses = [], []
    for step in range(1, 4):
        eng.forward(img_t)
        losses.append(float(eng.loss_backward(lab_t, 0.1)))
        eng.optimizer_step(lr=3e-3, weight_decay=1e-4, max_norm=1.0)
        tree = vit_ref.unflatten({k: x.astype(np.float32) for k, x in flat.items()})
        lr_, _, g = torch_ref.loss_and_grads(tree, images, labels, oc, 0.1)
        ref_losses.append(lr_)
        gn = vit_ref.global_norm(g.values())
        clip = 1.0 if gn < 1.0 else 1.0 / gn
        for k in flat:
            flat[k], m[k], v[k] = vit_ref.adamw_update(flat[k], g[k[len("params/"):]].astype(np.float64), m[k], v[k], step, 3e-3, 1e-4, clip)
    assert losses[-1] < losses[0]
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 3e-2 * max(1.0, abs(b)), (losses, ref_losses)
    got = _flat(eng.param_tree()["params"])
    for k, x in flat.items():
        kk = k[len("params/"):]
        # Adam's sign-like first steps amplify bf16 gradient noise on near-zero gradients; compare the update direction
        assert rel(got[kk], x) < 6e-2, (kk, rel(got[kk], x))


# ------------------------------------------------------------------------------------------ CaiT
CAIT_CASES = {
    "tiny_cait": dict(kind="cait", num_layers=2, num_heads=2, embed_dim=128, patch=8, num_classes=16, img_size=32, num_layers_token_only=2,
                      stoch_depth_rate=0.1, layerscale_eps=1e-5),
    "xxs2": dict(kind="cait", num_layers=2, num_heads=4, embed_dim=192, patch=16, num_classes=1000, img_size=224, num_layers_token_only=2,
                 stoch_depth_rate=0.05, layerscale_eps=1e-5),
    # the cait_m_* geometry: 16 heads of 48 (d = 768), 196 patches
    "m1": dict(kind="cait", num_layers=1, num_heads=16, embed_dim=768, patch=16, num_classes=1000, img_size=224, num_layers_token_only=1,
               stoch_depth_rate=0.2, layerscale_eps=1e-5),
}


@pytest.fixture(params=["materialising", "fused"])
def th_path(request, monkeypatch):
    """Both talking-heads paths of the CaiT engine: the default materialising kernels and the opt-in fused ones
    (CaiTEngine(th_fused=True), csrc/th_fused.hip; geometries they do not cover - 16 heads - keep the materialising kernels)."""
    monkeypatch.delenv("SAVIT_TH_FUSED", raising=False)
    return request.param


@pytest.mark.parametrize("case,B,training", [("tiny_cait", 32, False), ("tiny_cait", 32, True), ("xxs2", 2, True), ("m1", 2, True)])
def test_cait_forward_backward_parity(pkg, th_path, case, B, training):
    """Talking-heads SA + LayerScale + stochastic depth + class attention end to end vs the fp32 oracle / fp32 autograd.
    Training mode uses explicit per-sample keep masks (the JAX rng stream cannot be reproduced).  The tiny model runs 32 images: the
    gradient of its 2 x 2 talking-heads matrices is a strongly cancelling sum over (image, query, key) of bf16-rounded scores times
    cotangents, and with 3 images of 17 tokens its rounding noise alone was 4.7e-2 of the sum."""
    from savit_amd.cait_engine import CaiTEngine
    from savit_amd.config import ModelConfig

    kw = CAIT_CASES[case]
    mc, oc = ModelConfig(**kw), vit_ref.Cfg(**kw)
    rng = np.random.default_rng(21)
    params = vit_ref.init_params(oc, seed=6, randomize=True)
    images = vit_ref.bf16_round(rng.standard_normal((B, oc.img_size, oc.img_size, 3)).astype(np.float32))
    labels = rng.integers(0, oc.num_classes, B)
    masks = (rng.random((oc.num_layers + oc.num_layers_token_only, 2, B)) < 0.7).astype(np.float32) if training else None
    eng = CaiTEngine(mc, B, th_fused=(th_path == "fused"))
    assert eng.th_fused == (th_path == "fused" and mc.num_heads <= 8)
    eng.load_params(params)
    logits = eng.forward(torch.as_tensor(images).cuda(), is_training=training,
                         keep_masks=None if masks is None else torch.as_tensor(masks)).float().cpu().numpy()
    ref32 = vit_ref.forward(params, images, oc, mode="f32", is_training=training, keep_masks=masks)
    refbf = vit_ref.forward(params, images, oc, mode="bf16", is_training=training, keep_masks=masks)
    parity_bars.check_logits(f"cait:{case}:{training}", logits, ref32, refbf)
    loss = float(eng.loss_backward(torch.as_tensor(labels).cuda(), 0.1))
    loss_ref, _, grads_ref = torch_ref.loss_and_grads(params, images, labels, oc, 0.1, is_training=training, keep_masks=masks)
    assert abs(loss - loss_ref) < 2e-2 * max(1.0, abs(loss_ref))
    got = _flat(eng.grad_tree()["params"])
    assert set(got) == set(grads_ref)
    parity_bars.check_grads(f"cait:{case}:{training}", got, grads_ref)


def test_cait_reference_shapes_and_known_answers(pkg):
    """models/cait_test.py:13-40 (logits (2, 1000) on ones for the CaiT sizes; here the smallest and cait_s_24) + SURVEY 8c vi."""
    from savit_amd.model import create_model

    for name, count in (("cait_xxs_24", None), ("cait_s_24", 46_875_496), ("cait_m_24", None)):
        model = create_model(name, dtype=torch.bfloat16)
        logits, params = model.init_with_output(0, torch.ones(2, 224, 224, 3, device="cuda"), is_training=False)
        assert tuple(logits.shape) == (2, 1000)
        assert float(logits.float().abs().max()) == 0.0  # zero-init head (cait.py:179-182)
        p = params["params"]
        ls = p["Encoder_0"]["EncoderBlock_0"]["LayerScaleBlock_0"]["layerscale"]
        assert torch.all(ls == model.cfg.layerscale_eps)  # layerscale.py:5-10
        t = p["Encoder_0"]["EncoderBlock_0"]["SelfAttentionBlock_0"]["TalkingHeadsBlock_0"]["talking_heads_transform"]
        assert torch.allclose(t @ t.T, torch.eye(t.shape[0], device=t.device), atol=1e-5)  # orthogonal init
        assert "CAEncoderBlock_1" in p and "ClassSelfAttentionBlock_0" in p["CAEncoderBlock_0"]
        if count is not None:
            assert sum(v.numel() for _, v in torch_ref.leaves(p)) == count
        # training mode runs (stochastic depth masks drawn on the GPU) and differs from eval only through the masks
        model.bind(params)
        out_t = model(torch.ones(2, 224, 224, 3, device="cuda"), is_training=True)
        assert tuple(out_t.shape) == (2, 1000)


@pytest.mark.gpu
def test_backward_overlapped_wgrad_matches_single_stream():
    """The weight-gradient GEMMs run on a second stream (engine._Plan.run_overlapped); the gradients must equal the
    single-stream schedule's up to fp32 atomic summation order."""
    import torch
    from savit_amd.config import get_config
    from savit_amd.engine import ViTEngine

    cfg = get_config("vit_s_patch16", num_classes=104)
    B = 16
    eng = ViTEngine(cfg, B)
    eng.init_params(3)
    g = torch.Generator(device="cuda").manual_seed(5)
    eng.layout.view(eng.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes, device="cuda", generator=g) * 0.05)
    eng.weights_stale = True
    img = torch.randn(B, 224, 224, 3, device="cuda", generator=g).to(torch.bfloat16)
    lab = torch.randint(0, 104, (B,), device="cuda", generator=g, dtype=torch.int32)
    grads = {}
    for mode in (False, True, True):
        eng.overlap_wgrad = mode
        eng.forward(img)
        eng.loss_backward(lab)
        torch.cuda.synchronize()
        grads.setdefault(mode, []).append(eng.grads.clone())
    ref = grads[False][0]
    assert float(ref.abs().max()) > 0
    for got in grads[True]:
        err = float((got - ref).abs().max())
        assert err <= 1e-5 * max(1.0, float(ref.abs().max())), err


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["vit_ti_patch16", "cait_xxs_24"])
def test_flax_checkpoint_roundtrip(tmp_path, name):
    """Row f-4: a TrainState written in the Flax msgpack format restores into a fresh engine bit for bit (parameters, Adam
    moments, step) and carries the reference's module tree."""
    import torch
    from savit_amd import flax_ckpt
    from savit_amd.model import create_model

    torch.manual_seed(0)
    model = create_model(name, num_classes=16, dtype=torch.bfloat16)
    B = 2
    eng = model.engine(B)
    eng.init_params(11)
    eng.params.add_(0.01 * torch.randn_like(eng.params))  # non-trivial head / cls / LayerScale values
    eng.weights_stale = True
    img = torch.randn(B, 224, 224, 3, device="cuda").to(torch.bfloat16)
    lab = torch.randint(0, 16, (B,), device="cuda", dtype=torch.int32)
    eng.forward(img)
    eng.loss_backward(lab)
    eng.optimizer_step(1e-3, 1e-4, 1.0)
    ref_logits = eng.forward(img).clone()
    path = flax_ckpt.save_from_engine(eng, str(tmp_path), step=3)
    assert os.path.basename(path) == "checkpoint_3"
    state = flax_ckpt.read_train_state(path)
    assert set(state) == {"step", "params", "opt_state"} and set(state["opt_state"]) == {"0", "1", "2", "3"}
    assert "Encoder_0" in state["params"]["params"] and "mu" in state["opt_state"]["1"]

    model2 = create_model(name, num_classes=16, dtype=torch.bfloat16)
    eng2 = model2.engine(B)
    eng2.init_params(99)
    step = flax_ckpt.load_into_engine(eng2, state)
    eng2.weights_stale = True
    assert step == 3 and eng2.step_count == eng.step_count
    def leaves(t):
        return [x for v in t.values() for x in leaves(v)] if isinstance(t, dict) else [t]

    # (the flat buffers also hold alignment padding between tensors, which is not part of the tree: compare leaf by leaf)
    for a_flat, b_flat in ((eng.params, eng2.params), (eng.adam_m, eng2.adam_m), (eng.adam_v, eng2.adam_v)):
        la, lb = leaves(eng.layout.flax_tree(a_flat)), leaves(eng2.layout.flax_tree(b_flat))
        assert len(la) == len(lb) and all(torch.equal(x, y) for x, y in zip(la, lb))
    assert torch.equal(eng2.forward(img), ref_logits)


@pytest.mark.parametrize("name,B", [("vit_s_patch16", 6), ("vit_b_patch16", 4)])
def test_grouped_and_per_weight_weight_gradients_agree(name, B, monkeypatch):
    """Engine level: the tile FIFO of grouped launches (256 x 384 / 384 x 256 tiles for d = 384, 256 x 256 for d = 768; with and without a
    bound on the reach-back, with CUs reserved for a resident all-reduce) against one launch per weight - same gradients up to fp32
    summation order, same loss."""
    from savit_amd.config import get_config
    from savit_amd.engine import ViTEngine

    cfg = get_config(name)
    g = torch.Generator(device="cuda").manual_seed(3)
    img = torch.randn(B, 224, 224, 3, device="cuda", generator=g).to(torch.bfloat16)
    lab = torch.randint(0, 1000, (B,), device="cuda", generator=g, dtype=torch.int32)

    def grads(**kw):
        eng = ViTEngine(cfg, B, **kw)
        eng.init_params(5)
        eng.layout.view(eng.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes, generator=torch.Generator().manual_seed(1)) * 0.03)
        eng.forward(img)
        eng.loss_backward(lab)
        torch.cuda.synchronize()
        return eng, eng.grads.clone(), float(eng.loss.item())

    monkeypatch.setenv("SAVIT_WGRAD_GROUP", "0")
    e0, g0, l0 = grads()
    assert e0.wgrad_tile == 0
    monkeypatch.delenv("SAVIT_WGRAD_GROUP")
    e1, g1, l1 = grads()
    assert e1.wgrad_tile == (640 if cfg.embed_dim == 384 else 256)
    e2, g2, l2 = grads(wgrad_max_lag=2, reserved_cus=32)
    assert e2.wgrad_lag <= 2 and e2.wgrad_cap == e2.n_cus - 32
    assert abs(l0 - l1) < 2e-6 * l0 and abs(l0 - l2) < 2e-6 * l0  # same forward; the scalar loss is an fp32 atomic sum over the rows
    for gg in (g1, g2):
        for nm in ("l0.Wqkv", "l3.W1", f"l{cfg.num_layers - 1}.W2", "l5.Wo", "Wpe", "pos"):
            a, b = e0.layout.view(g0, nm), e0.layout.view(gg, nm)
            assert float((a - b).norm() / a.norm()) < 2e-6, nm
        assert float((g0 - gg).norm() / g0.norm()) < 2e-6
    g1b = grads()[1]  # and the grouped form is bitwise repeatable (the per-weight launches of this small batch use fp32 atomics)
    for l in range(cfg.num_layers):
        for w in ("Wqkv", "Wo", "W1", "W2"):
            if not (w == "Wo" and l in e1.wgrad_divert):
                assert torch.equal(e0.layout.view(g1, f"l{l}.{w}"), e0.layout.view(g1b, f"l{l}.{w}")), (l, w)


@pytest.mark.parametrize("name,B", [("vit_b_patch16", 4), ("vit_s_patch16", 6)])
def test_first_touch_gradients_and_folded_norm(name, B, monkeypatch):
    """Round 5: grouped weight gradients stored by first touch (no memset of the gradient buffer), the LayerNorm
    column sums reduced by one launch at the end of backward, the gradient norm's sum of squares carried by the weight-gradient
    launches.  Against the same engine with all three turned off: identical weight gradients bit for bit (same kernels, same order),
    the column sums within fp32 atomic order, the same clipped update; gradient accumulation (zero_grads=False) still accumulates; an
    engine with DDP hooks attached keeps the per-launch forms."""
    from savit_amd.config import get_config
    from savit_amd.engine import ViTEngine

    cfg = get_config(name)
    g = torch.Generator(device="cuda").manual_seed(3)
    img = torch.randn(B, 224, 224, 3, device="cuda", generator=g).to(torch.bfloat16)
    lab = torch.randint(0, 1000, (B,), device="cuda", generator=g, dtype=torch.int32)

    def make():
        eng = ViTEngine(cfg, B)
        eng.init_params(5)
        eng.layout.view(eng.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes, generator=torch.Generator().manual_seed(1)) * 0.03)
        return eng

    def run(eng, poison=True):
        if poison:
            eng.grads.fill_(float("nan"))  # whatever the buffer held before must not matter
        eng.forward(img)
        eng.loss_backward(lab)
        torch.cuda.synchronize()
        return eng.grads.clone()

    monkeypatch.setenv("SAVIT_WGRAD_FIRST_TOUCH", "0")
    monkeypatch.setenv("SAVIT_DEFER_LN_FINALIZE", "0")
    e0 = make()
    g0 = run(e0)
    labels0 = [c[2] for c in e0._serial_bwd_plan().calls]
    monkeypatch.delenv("SAVIT_WGRAD_FIRST_TOUCH")
    monkeypatch.delenv("SAVIT_DEFER_LN_FINALIZE")
    e1 = make()
    g1 = run(e1)
    plan = e1._serial_bwd_plan()
    assert plan.rest_ranges is not None and plan.fold_sumsq and 0 < plan.rest_elems < 0.10 * e1.grads.numel()  # (biases, LayerNorm parameters, embeddings, head, and the last layer's three 128-row weight gradients)
    assert sum(1 for c in plan.calls if c[2] == "ln.bwd.finalize") == 1 and len(plan.calls) < len(labels0) + 2
    assert torch.isfinite(g1).all()
    lay = e1.layout
    for l in range(cfg.num_layers):
        for w in ("Wqkv", "Wo", "W1", "W2"):
            assert torch.equal(lay.view(g0, f"l{l}.{w}"), lay.view(g1, f"l{l}.{w}")), (l, w)
        for v in ("ln1_g", "ln1_b", "ln2_g", "ln2_b", "b1", "b2"):
            a, b = lay.view(g0, f"l{l}.{v}"), lay.view(g1, f"l{l}.{v}")
            assert float((a - b).norm() / a.norm().clamp_min(1e-20)) < 2e-6, (l, v)
    for nm in ("Wpe", "pos", "cls", "Wh", "bh", "lnf_g"):
        a, b = lay.view(g0, nm), lay.view(g1, nm)
        assert float((a - b).norm() / a.norm().clamp_min(1e-20)) < 2e-6, nm
    # the folded norm is the norm: same clipped AdamW update as the engine that sums squares over the whole buffer
    ss_full = float((g1.double() ** 2).sum())
    for e in (e0, e1):
        e.optimizer_step(lr=1e-3, weight_decay=1e-4, max_norm=0.5)
    torch.cuda.synchronize()
    assert abs(float(e1.gnorm[0]) - ss_full) < 1e-5 * ss_full and abs(float(e0.gnorm[0]) - ss_full) < 1e-5 * ss_full
    assert float((e0.params - e1.params).abs().max()) < 2e-6
    # second step from the updated parameters: nothing stale is left in the accumulators
    g1b = run(e1, poison=False)
    ss2 = float((g1b.double() ** 2).sum())
    e1.optimizer_step(lr=1e-3, weight_decay=1e-4, max_norm=0.5)
    torch.cuda.synchronize()
    assert abs(float(e1.gnorm[0]) - ss2) < 1e-5 * ss2
    # gradient accumulation: two backward passes without clearing add up (first touch is off for the second)
    e2 = make()
    ga = run(e2)
    e2.forward(img)
    e2.loss_backward(lab, zero_grads=False)
    torch.cuda.synchronize()
    assert float((e2.grads - 2 * ga).norm() / (2 * ga).norm()) < 2e-6
    e2.optimizer_step(lr=1e-3, max_norm=0.5)  # the accumulators are not the norm of an accumulated gradient: whole-buffer pass
    torch.cuda.synchronize()
    ss3 = float((e2.grads.double() ** 2).sum())
    assert abs(float(e2.gnorm[0]) - ss3) < 1e-5 * ss3
    gc = run(e2, poison=False)  # and the next ordinary pass stores by first touch again
    assert float((gc - run(make())).norm()) >= 0.0 and torch.isfinite(gc).all()
    # a data-parallel rank (hooks attached): per-launch finalizes, whole-buffer norm, first touch kept
    e3 = make()
    fired = []
    e3.bwd_hooks = {f"l{cfg.num_layers // 2}.ln1.bwd": lambda: fired.append(1), "Wpe.wgrad": lambda: fired.append(2)}
    g3 = run(e3)
    p3 = e3._serial_bwd_plan()
    assert fired == [1, 2] and not p3.fold_sumsq and not any(c[2] == "ln.bwd.finalize" for c in p3.calls) and p3.rest_ranges is not None
    assert float((g3 - g1).norm() / g1.norm()) < 2e-6
    # opt-in: a cls-only weight that found no slot in the tile FIFO as ONE grouped launch of its own (instead of the split pair + reduce)
    monkeypatch.setenv("SAVIT_WGRAD_SMALL_GROUPS", "1")
    e4 = make()
    g4 = run(e4)
    own = [c[2] for c in e4._serial_bwd_plan().calls if c[2].endswith(".wgrad") and c[2].startswith(f"l{cfg.num_layers - 1}.")]
    assert all((lbl + ".reduce") not in [c[2] for c in e4._serial_bwd_plan().calls] for lbl in own)
    assert torch.isfinite(g4).all() and float((g4 - g1).norm() / g1.norm()) < 2e-6


@pytest.mark.parametrize("name,B,img", [("vit_b_patch16", 4, 224), ("vit_s_patch16", 6, 224), ("vit_ti_patch16", 3, 224)])
def test_last_layer_backward_on_cls_rows_only(name, B, img, monkeypatch):
    """Round 5: behind the final LayerNorm only the cls rows carry a gradient (vit.py:57,95), so the last encoder layer's MLP branch,
    second LayerNorm and output projection are differentiated on B rows instead of B*N - the rows left out contribute exact zeros.
    Against the dense plan (SAVIT_CLS_ONLY_LAST=0): every gradient that does not depend on a changed summation order bit for bit
    (all layers below the last, and the last layer's qkv), the last layer's MLP / projection weights within fp32 summation order (a
    128-row product per weight instead of tiles of the grouped launch); no zero-fill of the [B*N, d] residual-gradient buffers."""
    from savit_amd.config import get_config
    from savit_amd.engine import ViTEngine

    cfg = get_config(name, img_size=img)
    g = torch.Generator(device="cuda").manual_seed(9)
    imgs = torch.randn(B, img, img, 3, device="cuda", generator=g).to(torch.bfloat16)
    lab = torch.randint(0, 1000, (B,), device="cuda", generator=g, dtype=torch.int32)

    def run(only_last, fwd, rows_tile=True):
        # rows_tile True (the product default): the B-row products on the few-rows kernel (tile 24).  One accumulator per output element
        # walks K in ascending 32-steps there as in every LDS tile (csrc/gemm_tn.hip, test_gemm_few_rows_kernel_is_bitwise_the_lds_tiles), so
        # the cls-row BACKWARD is held to the fp32-summation-order bar against the dense plan WITH the kernel the product uses.  (Round 5
        # forced the LDS tiles here: a workaround for the kernel's first, K-splitting form - gpurun_out/r5y/tests.log - that outlived it;
        # VERDICT r5 weak 2.)  rows_tile False is kept as a third run: same bar.
        eng = ViTEngine(cfg, B, cls_only_last=only_last, cls_fwd=fwd, rows_tile=rows_tile)
        eng.init_params(5)
        eng.layout.view(eng.params, "Wh").copy_(torch.randn(cfg.embed_dim, cfg.num_classes, generator=torch.Generator().manual_seed(1)) * 0.03)
        eng.dres.fill_(float("nan"))  # stale contents of the dense residual-gradient buffers must not matter
        for t in eng.dres_b_ring:
            t.fill_(float("nan"))
        eng.forward(imgs)
        eng.loss_backward(lab)
        torch.cuda.synchronize()
        return eng, eng.grads.clone(), [c[2] for c in eng._serial_bwd_plan().calls], [c[2] for c in eng._fwd_plan.calls]

    e0, g0, _, fl0 = run(False, True)
    assert not e0.cls_only_last and not e0.cls_fwd
    e1, g1, labels1, fl1 = run(True, False)
    assert e1.opt.rows_tile and e1.cls_only_last and not e1.cls_fwd and "zero.d_o" in labels1 and torch.isfinite(g1).all() and fl1 == fl0
    e1b, g1b, _, _ = run(True, False, rows_tile=False)
    e2, g2, labels2, fl2 = run(True, True)
    assert e2.cls_fwd == (cfg.head_dim in (48, 64) and cfg.seq_len <= 640) and torch.isfinite(g2).all()
    lay, NL = e1.layout, cfg.num_layers
    assert abs(float(e0.loss) - float(e1.loss)) < 2e-6 * float(e0.loss)  # same forward; the scalar loss is an fp32 atomic sum over the rows
    worst = (0.0, "")
    for tag, gg in (("rows kernel", g1), ("LDS tiles", g1b)):
        for l in range(NL):
            for v in ("Wqkv", "Wo", "W1", "W2", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "b1", "b2"):
                a, b = lay.view(g0, f"l{l}.{v}"), lay.view(gg, f"l{l}.{v}")
                r = float((a - b).norm() / a.norm().clamp_min(1e-20))
                worst = max(worst, (r, f"{tag} l{l}.{v}"))
                assert r < 3e-6, (tag, l, v, r)
        for nm in ("Wpe", "pos", "cls", "Wh", "bh", "lnf_g", "lnf_b"):
            a, b = lay.view(g0, nm), lay.view(gg, nm)
            r = float((a - b).norm() / a.norm().clamp_min(1e-20))
            worst = max(worst, (r, f"{tag} {nm}"))
            assert r < 3e-6, (tag, nm, r)
    print("cls-row backward vs dense plan: worst relative gradient difference", worst)
    # the few-rows kernel and the LDS tiles are the same function bit for bit: so is everything below the B-row products in the two
    # cls-row runs - every weight-gradient matrix the grouped launches store (fixed summation order; the bias / LayerNorm column sums
    # and small per-weight launches meet in fp32 atomics and are held to the bar above)
    for l in range(NL - 1):
        for v in ("Wqkv", "Wo", "W1", "W2"):
            if not (v == "Wo" and l in e1.wgrad_divert) and e1.wgrad_tile:
                assert torch.equal(lay.view(g1, f"l{l}.{v}"), lay.view(g1b, f"l{l}.{v}")), (l, v)
    # the cotangent entering the layers below is the same bit for bit where the arithmetic is the same: qkv of the last layer
    assert torch.equal(lay.view(g0, f"l{NL - 1}.Wqkv"), lay.view(g1, f"l{NL - 1}.Wqkv")) or cfg.embed_dim % 256 != 0
    if not e2.cls_fwd:
        return
    # forward on the cls rows too: the same rounding points (bf16 P / O / activations), other fp32 summation orders inside the cls
    # query's attention and the B-row products, so bf16 outputs may differ in their last bit - the bars are those of two bf16 runs
    assert any(c[0] is e2.L.savit_cls_query_attention_fwd for c in e2._fwd_plan.calls) and fl2 == fl0
    assert any(c[0] is e2.L.savit_cls_query_attention_bwd for c in e2._serial_bwd_plan().calls)
    assert abs(float(e0.loss) - float(e2.loss)) < 2e-3 * float(e0.loss)
    worst = 0.0
    for nm in lay.off:
        a, b = lay.view(g0, nm), lay.view(g2, nm)
        r = float((a - b).norm() / a.norm().clamp_min(1e-20))
        worst = max(worst, r)
        assert r < 2e-2, (nm, r)
    print("cls forward vs dense: worst relative gradient difference", worst)
