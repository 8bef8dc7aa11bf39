"""Pins the CPU oracle (oracle/vit_ref.py): known answers derivable from the reference source
(SURVEY.md 8c i-viii), agreement with the independent torch composition, golden fixtures.
CPU only."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref, vit_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")

TINY_VIT = vit_ref.Cfg(kind="vit", num_layers=2, num_heads=2, embed_dim=32, patch=8, num_classes=10, img_size=32)
TINY_CAIT = vit_ref.Cfg(kind="cait", num_layers=2, num_heads=2, embed_dim=32, patch=8, num_classes=10, img_size=32,
                        num_layers_token_only=2, stoch_depth_rate=0.1, layerscale_eps=1e-5)


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# ---- (i) zero-init head => logits == 0, loss == ln(1000)  (vit.py:96-98, train.py:83-90)
def test_zero_head_known_answer():
    cfg = vit_ref.get_cfg("vit_ti_patch16")
    params = vit_ref.init_params(cfg, seed=0)  # reference initialisers: zero head kernel/bias
    x = np.random.default_rng(0).standard_normal((2, 224, 224, 3)).astype(np.float32)
    logits = vit_ref.forward(params, x, cfg)
    assert logits.shape == (2, 1000)
    assert np.all(logits == 0.0)
    loss = vit_ref.loss_fn(logits, np.array([3, 7]), label_smoothing=0.1)
    assert abs(loss - math.log(1000.0)) < 1e-5  # 6.907755
    # dL/dbias = softmax - y_smooth = 1/1000 - y_smooth (SURVEY 8c i)
    g = vit_ref.dloss_dlogits(logits, np.array([3, 7]), 0.1) * 2
    assert abs(g[0, 3] - (1e-3 - 0.9001)) < 1e-6 and abs(g[0, 0] - (1e-3 - 1e-4)) < 1e-7


# ---- (ii) logits shape on ones input (vit_test.py:13-26) incl. N=50 for /32 and 197 for /16
@pytest.mark.parametrize("name,N", [("vit_b_patch32", 50), ("vit_b_patch16", 197)])
def test_logits_shape_reference_tests(name, N):
    cfg = vit_ref.get_cfg(name)
    assert cfg.seq_len == N
    if name == "vit_b_patch16":  # shape logic is identical; keep the CPU suite fast
        cfg = vit_ref.Cfg(**{**vit_ref.MODEL_ZOO[name], "num_layers": 1})
    params = vit_ref.init_params(cfg, seed=0)
    logits = vit_ref.forward(params, np.ones((2, 224, 224, 3), np.float32), cfg, is_training=True)
    assert logits.shape == (2, 1000)


# ---- (iii) parameter counts from the module definitions
@pytest.mark.parametrize("name,img,count", [
    ("vit_ti_patch16", 224, 5_708_008), ("vit_s_patch16", 224, 22_031_848), ("vit_b_patch16", 224, 86_530_024),
    ("vit_l_patch16", 384, 304_616_424), ("cait_s_24", 224, 46_875_496)])
def test_param_counts(name, img, count):
    cfg = vit_ref.get_cfg(name, img_size=img)
    shapes = vit_ref.param_shapes(cfg)
    assert sum(int(np.prod(s)) for s in shapes.values()) == count


@pytest.mark.parametrize("cfg", [TINY_VIT, TINY_CAIT])
def test_param_shapes_match_init(cfg):
    flat = vit_ref.flatten(vit_ref.init_params(cfg, 0))
    shapes = vit_ref.param_shapes(cfg)
    assert set(flat) == set(shapes)
    for k, v in flat.items():
        assert tuple(v.shape) == shapes[k], k
        assert v.dtype == np.float32


def test_unknown_model_raises():  # create_model.py:214-215
    with pytest.raises(RuntimeError, match="Model not found."):
        vit_ref.get_cfg("resnet50")


# ---- (iv) is_training does not change ViT output; (v) CaiT eval deterministic, CA output shape
def test_is_training_invariance_vit_and_cait_eval():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 32, 32, 3)).astype(np.float32)
    pv = vit_ref.init_params(TINY_VIT, 1, randomize=True)
    assert np.array_equal(vit_ref.forward(pv, x, TINY_VIT, is_training=True), vit_ref.forward(pv, x, TINY_VIT, is_training=False))
    pc = vit_ref.init_params(TINY_CAIT, 1, randomize=True)
    a = vit_ref.forward(pc, x, TINY_CAIT, is_training=False)
    assert np.array_equal(a, vit_ref.forward(pc, x, TINY_CAIT, is_training=False))
    # all-ones keep masks with scale_by_keep => training output = eval scaled branches differ
    masks = np.ones((4, 2, 2), np.float32)
    b = vit_ref.forward(pc, x, TINY_CAIT, is_training=True, keep_masks=masks)
    assert b.shape == a.shape and not np.allclose(a, b)


def test_class_attention_output_shape():  # cait.py:14-15
    pol = vit_ref.Policy("f32")
    p = vit_ref.init_params(TINY_CAIT, 0)["params"]["CAEncoderBlock_0"]["ClassSelfAttentionBlock_0"]
    x = np.random.default_rng(0).standard_normal((3, 17, 32)).astype(np.float32)
    assert vit_ref.attention_block(pol, p, x[:, 0:1], x, 2).shape == (3, 1, 32)


# ---- (vi) LayerScale init == eps exactly; (vii) softmax rows sum to 1; (viii) smooth_labels
def test_small_known_answers():
    cfg = vit_ref.get_cfg("cait_xxs_24")
    shapes_only = vit_ref.Cfg(**{**vit_ref.MODEL_ZOO["cait_xxs_24"], "num_layers": 1})
    p = vit_ref.init_params(shapes_only, 0)["params"]
    assert np.all(p["Encoder_0"]["EncoderBlock_0"]["LayerScaleBlock_0"]["layerscale"] == np.float32(cfg.layerscale_eps))
    assert np.all(p["cls"] == 0) and np.all(p["Dense_0"]["kernel"] == 0)
    t = p["Encoder_0"]["EncoderBlock_0"]["SelfAttentionBlock_0"]["TalkingHeadsBlock_0"]["talking_heads_transform"]
    assert np.allclose(t @ t.T, np.eye(t.shape[0]), atol=1e-5)  # orthogonal init (talking_heads.py:12)
    s = vit_ref.softmax_last(vit_ref.Policy("f32"), np.random.default_rng(0).standard_normal((4, 7, 9)).astype(np.float32) * 5)
    assert np.allclose(s.sum(-1), 1.0, atol=1e-6)
    y = vit_ref.smooth_labels(vit_ref.one_hot(np.array([2]), 1000), 0.1)
    assert abs(y[0, 2] - 0.9001) < 1e-7 and abs(y[0, 0] - 0.0001) < 1e-9
    # identical patch tokens before pos-embed on a ones image (SURVEY 8c vii)
    tok = vit_ref.patchify(np.ones((1, 32, 32, 3), np.float32), 8, 8)
    assert np.all(tok == tok[:, :1])


def test_patchify_order():  # (ph pw c), c fastest; patches row-major over (h, w)
    img = np.arange(1 * 4 * 4 * 3, dtype=np.float32).reshape(1, 4, 4, 3)
    t = vit_ref.patchify(img, 2, 2)
    assert t.shape == (1, 4, 12)
    # patch (h=0,w=1): pixels (0,2),(0,3),(1,2),(1,3), channels fastest
    expect = np.concatenate([img[0, 0, 2], img[0, 0, 3], img[0, 1, 2], img[0, 1, 3]])
    assert np.array_equal(t[0, 1], expect)


def test_layernorm_eps_and_gelu_hazards():
    """The two parity hazards of SURVEY section 0: eps 1e-6 (not 1e-5) and tanh-GELU (not erf)."""
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((5, 768)) * 1e-2).astype(np.float32)
    ours = vit_ref.layer_norm(vit_ref.Policy("f32"), x, np.ones(768, np.float32), np.zeros(768, np.float32))
    t6 = torch.nn.functional.layer_norm(torch.tensor(x), (768,), eps=1e-6).numpy()
    t5 = torch.nn.functional.layer_norm(torch.tensor(x), (768,), eps=1e-5).numpy()
    assert rel(ours, t6) < 1e-5 and rel(ours, t5) > 1e-3
    g = vit_ref.gelu_tanh(vit_ref.Policy("f32"), x * 100)
    assert rel(g, torch.nn.functional.gelu(torch.tensor(x * 100), approximate="tanh").numpy()) < 1e-6
    assert np.abs(g - torch.nn.functional.gelu(torch.tensor(x * 100)).numpy()).max() > 1e-4


# ---- independent torch composition agrees (fp32 <= 1e-6 rel; fp64 <= 1e-12)
@pytest.mark.parametrize("cfg", [TINY_VIT, TINY_CAIT], ids=["vit", "cait"])
def test_two_restatements_agree(cfg):
    rng = np.random.default_rng(7)
    params = vit_ref.init_params(cfg, 7, randomize=True)
    x = rng.standard_normal((3, 32, 32, 3)).astype(np.float32)
    for mode, dt, tol in (("f32", torch.float32, 2e-6), ("f64", torch.float64, 1e-12)):
        a = vit_ref.forward(params, x, cfg, mode=mode)
        b = torch_ref.forward(torch_ref.to_torch(params["params"], dt), torch.as_tensor(x).to(dt), cfg).numpy()
        assert rel(a, b) < tol, (mode, rel(a, b))
    if cfg.kind == "cait":
        masks = (rng.random((4, 2, 3)) < 0.6).astype(np.float32)
        a = vit_ref.forward(params, x, cfg, mode="f64", is_training=True, keep_masks=masks)
        b = torch_ref.forward(torch_ref.to_torch(params["params"], torch.float64), torch.as_tensor(x).double(), cfg,
                              True, torch.as_tensor(masks).double()).numpy()
        assert rel(a, b) < 1e-12


def test_real_width_block_agrees():
    """One encoder block at DeiT-S width (d=384, H=6, N=197) - both restatements, fp32."""
    cfg = vit_ref.Cfg(**{**vit_ref.MODEL_ZOO["vit_s_patch16"], "num_layers": 1})
    params = vit_ref.init_params(cfg, 3, randomize=True)
    x = np.random.default_rng(3).standard_normal((2, 224, 224, 3)).astype(np.float32)
    a = vit_ref.forward(params, x, cfg, mode="f32")
    b = torch_ref.forward(torch_ref.to_torch(params["params"]), torch.as_tensor(x), cfg).numpy()
    assert rel(a, b) < 5e-6


def test_analytic_dlogits_matches_autograd():
    rng = np.random.default_rng(0)
    logits = rng.standard_normal((6, 10))
    labels = rng.integers(0, 10, 6)
    t = torch.tensor(logits, requires_grad=True)
    torch_ref.loss_from_logits(t, torch.as_tensor(labels), 0.1).backward()
    assert rel(vit_ref.dloss_dlogits(logits, labels, 0.1), t.grad.numpy()) < 1e-12
    assert abs(vit_ref.loss_fn(logits, labels, 0.1) - float(torch_ref.loss_from_logits(torch.tensor(logits), torch.as_tensor(labels), 0.1))) < 1e-12


def test_adamw_matches_torch():
    """optax chain: additive weight decay folded into the update BEFORE the lr scale
    == torch.optim.AdamW only when decoupled decay is lr*wd*p; check against a hand composition."""
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal(50)
    g = rng.standard_normal(50)
    p, m, v = p0.copy(), np.zeros(50), np.zeros(50)
    tp = torch.tensor(p0.copy(), requires_grad=True)
    opt = torch.optim.AdamW([tp], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    for step in range(1, 4):
        p, m, v = vit_ref.adamw_update(p, g, m, v, step, 3e-3, 1e-4)
        tp.grad = torch.tensor(g.copy())
        opt.step()
    # torch AdamW: p *= (1-lr*wd) then p -= lr*mhat/(sqrt(vhat)+eps): differs from optax by O(lr^2 wd)
    assert rel(p, tp.detach().numpy()) < 1e-6


def test_topk_and_schedule():
    logits = np.array([[0.1, 0.9, 0.3, 0.2, 0.0, -1.0], [5, 4, 3, 2, 1, 0.5]], np.float32)
    r = vit_ref.topk_correct(logits, np.array([1, 5]))
    assert r["top_1_acc"].tolist() == [1.0, 0.0] and r["top_5_acc"].tolist() == [1.0, 0.0]
    assert vit_ref.warmup_cosine_lr(0, 1.0, 10, 100) == 0.0
    assert abs(vit_ref.warmup_cosine_lr(10, 1.0, 10, 100) - 1.0) < 1e-12
    assert abs(vit_ref.warmup_cosine_lr(100, 1.0, 10, 100) - 1e-5) < 1e-12


def test_bf16_round():
    x = np.array([1.0, 1.00390625, 1.0 + 2 ** -9, 3.14159, -2.5e-5, 65504.0], np.float32)
    assert np.array_equal(vit_ref.bf16_round(x), torch.tensor(x).bfloat16().float().numpy())
    big = np.random.default_rng(0).standard_normal(10000).astype(np.float32)
    assert np.array_equal(vit_ref.bf16_round(big), torch.tensor(big).bfloat16().float().numpy())


# ---- golden fixtures
TINY_MIXER = vit_ref.Cfg(kind="mixer", num_layers=2, num_heads=1, embed_dim=32, patch=8, num_classes=10, img_size=32)
TINY_TNT = vit_ref.Cfg(kind="tnt", num_layers=2, num_heads=2, embed_dim=32, patch=16, num_classes=10, img_size=32, inner_num_heads=2,
                       inner_embed_dim=8)


@pytest.mark.parametrize("cfg,name", [(TINY_VIT, "tiny_vit.npz"), (TINY_CAIT, "tiny_cait.npz"), (TINY_MIXER, "tiny_mixer.npz"),
                                      (TINY_TNT, "tiny_tnt.npz")])
def test_golden(cfg, name):
    z = np.load(os.path.join(GOLD, name))
    params = vit_ref.unflatten({k[2:]: z[k] for k in z.files if k.startswith("P:")})
    logits = vit_ref.forward(params, z["images"], cfg, mode="f64")
    assert rel(logits, z["logits"]) < 1e-6
    assert abs(vit_ref.loss_fn(logits, z["labels"], 0.1) - float(z["loss"])) < 1e-5
    assert rel(vit_ref.forward(params, z["images"], cfg, mode="bf16"), z["logits_bf16"]) < 1e-6
    _, _, grads = torch_ref.loss_and_grads(params, z["images"], z["labels"], cfg, 0.1)
    for k, g in grads.items():
        ref = z["G:params/" + k]
        assert np.abs(g - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k
    if cfg.kind == "cait":
        lt = vit_ref.forward(params, z["images"], cfg, mode="f64", is_training=True, keep_masks=z["keep_masks"])
        assert rel(lt, z["logits_train"]) < 1e-6


def test_bf16_emulation_noise_floor():
    """Documents SURVEY A.5: the reference's own bf16 graph differs from fp32 math by O(1e-2) on
    logits, so the 1e-3 target applies per kernel, not end to end."""
    z = np.load(os.path.join(GOLD, "tiny_vit.npz"))
    r = rel(z["logits_bf16"], z["logits"])
    assert 1e-4 < r < 5e-2


@pytest.mark.parametrize("name", ["block_d192_n197", "block_d384_n197", "block_cait_d384_n196", "block_d768_n197", "block2_d384_n197",
                                  "block2_d768_n197"])
def test_block_fixtures_pin_the_oracle(name):
    """Real-width single-block fixtures (tests/golden/block_*.npz): the inputs regenerated from the seed match the stored checksums
    and the oracle reproduces the stored fp64 logits, bf16-emulated logits and loss (the gradients too for the smallest block; the
    d 1024 / N 577 fixture is exercised by the GPU test only - its fp64 forward takes a minute on this container's cores)."""
    from tests.golden import make_golden

    fx = np.load(os.path.join(GOLD, name + ".npz"))
    kw, seed = make_golden.BLOCKS[name]
    cfg, params, images, labels = make_golden.block_inputs(kw, seed)
    for k, v in make_golden.checksums(params, images).items():
        np.testing.assert_allclose(v, fx[k], rtol=1e-12, atol=1e-12, err_msg=k)
    assert np.array_equal(labels, fx["labels"])
    logits = vit_ref.forward(params, images, cfg, mode="f64")
    assert rel(logits, fx["logits"]) < 1e-9
    assert abs(vit_ref.loss_fn(logits, labels, 0.1) - float(fx["loss"])) < 1e-9
    assert rel(vit_ref.forward(params, images, cfg, mode="bf16"), fx["logits_bf16"]) < 1e-6
    assert rel(vit_ref.forward(params, images, cfg, mode="engine"), fx["logits_engine"]) < 1e-6  # the engine-rounding policy (round 3)
    if name == "block_d192_n197":
        _, _, grads = torch_ref.loss_and_grads(params, images, labels, cfg, 0.1, dtype=torch.float64)
        for k, g in grads.items():
            g = np.asarray(g, np.float64).ravel()
            assert abs(np.linalg.norm(g) - float(fx["GN:" + k])) <= 1e-9 * max(1.0, float(fx["GN:" + k])), k
            assert rel(g[fx["GI:" + k]], fx["GV:" + k]) < 1e-6, k
