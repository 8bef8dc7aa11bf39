"""Pins the MLP-Mixer part of the CPU oracle (oracle/vit_ref.py mixer_forward; /root/reference/models/mlp_mixer.py) and the
host-side layout of the HIP engine (padded token-mixing parameters).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import torch_ref, vit_ref

TINY = vit_ref.Cfg(kind="mixer", num_layers=2, num_heads=1, embed_dim=32, patch=8, num_classes=10, img_size=32)


def _closed_form_count(L, d, patch, n, C=1000):
    """Parameter count from the module definitions: Dense(use_bias=True) everywhere (patch_embed.py:23-25 with use_bias=True,
    ff.py:26-31, mlp_mixer.py:63), two LayerNorms per block + one final, token FF hidden int(0.5 n), channel FF hidden 4 d."""
    ft, f = max(1, int(0.5 * n)), 4 * d
    block = 2 * (2 * d) + (n * ft + ft + ft * n + n) + (d * f + f + f * d + d)
    return (patch * patch * 3 * d + d) + L * block + 2 * d + d * C + C


@pytest.mark.parametrize("name,L,d,patch", [("mixer_s_patch32", 8, 512, 32), ("mixer_s_patch16", 8, 512, 16), ("mixer_b_patch32", 12, 768, 32),
                                            ("mixer_b_patch16", 12, 768, 16), ("mixer_l_patch32", 24, 1024, 32), ("mixer_l_patch16", 32, 1024, 16)])
def test_param_counts_and_geometry(name, L, d, patch):  # create_model.py:184-213
    cfg = vit_ref.get_cfg(name)
    assert (cfg.kind, cfg.num_layers, cfg.embed_dim, cfg.patch) == ("mixer", L, d, patch)
    n = (224 // patch) ** 2
    assert cfg.n_patches == n and cfg.tokens_hidden == int(0.5 * n) and cfg.hidden == 4 * d
    shapes = vit_ref.param_shapes(cfg)
    assert sum(int(np.prod(s)) for s in shapes.values()) == _closed_form_count(L, d, patch, n)
    assert shapes[f"params/MixerBlock_{L - 1}/FFBlock_0/Dense_0/kernel"] == (n, int(0.5 * n))
    assert shapes[f"params/MixerBlock_0/FFBlock_0/Dense_1/bias"] == (n,)
    assert shapes["params/PatchEmbedBlock_0/Dense_0/bias"] == (d,)


def test_shapes_match_init_and_logits_shape():
    p = vit_ref.init_params(TINY, seed=0)
    assert {k: tuple(v.shape) for k, v in vit_ref.flatten(p).items()} == vit_ref.param_shapes(TINY)
    # flax defaults: zero biases, unit LayerNorm scale, NON-zero head kernel (mlp_mixer.py:63 keeps the default initialiser)
    q = p["params"]
    assert np.all(q["MixerBlock_0"]["FFBlock_0"]["Dense_1"]["bias"] == 0) and np.all(q["LayerNorm_0"]["scale"] == 1)
    assert np.abs(q["Dense_0"]["kernel"]).max() > 0
    x = np.ones((2, 32, 32, 3), np.float32)
    logits = vit_ref.forward(p, x, TINY)
    assert logits.shape == (2, 10)
    assert np.array_equal(logits, vit_ref.forward(p, x, TINY, is_training=True))  # every dropout rate is 0


def test_two_restatements_agree():
    p = vit_ref.init_params(TINY, seed=3, randomize=True)
    x = np.random.default_rng(1).standard_normal((3, 32, 32, 3)).astype(np.float32)
    a = vit_ref.forward(p, x, TINY, mode="f32")
    b = torch_ref.forward(torch_ref.to_torch(p["params"]), torch.as_tensor(x), TINY).numpy()
    assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max())
    c = vit_ref.forward(p, x, TINY, mode="f64")
    assert np.abs(a - c).max() < 1e-5


def test_token_mixing_mixes_tokens_only():
    """With the channel FF and every LayerNorm bias zeroed out of the picture, a permutation of the channels commutes with the
    token-mixing FF (it acts on the token axis alone): mlp_mixer.py:19-23."""
    pol = vit_ref.Policy("f64")
    rng = np.random.default_rng(0)
    n, d, ft = 16, 8, 8
    p = {"Dense_0": {"kernel": rng.standard_normal((n, ft)), "bias": rng.standard_normal(ft)},
         "Dense_1": {"kernel": rng.standard_normal((ft, n)), "bias": rng.standard_normal(n)}}
    x = rng.standard_normal((2, n, d))
    perm = rng.permutation(d)
    f = lambda t: np.swapaxes(vit_ref.ff_block(pol, p, np.swapaxes(t, -1, -2)), -1, -2)  # noqa: E731
    assert np.allclose(f(x)[..., perm], f(x[..., perm]), atol=1e-12)


def test_token_output_bias_has_zero_gradient():
    """Known answer: the second token-mixing Dense's bias adds one value per token to EVERY channel; LayerNorm over channels is
    invariant to such a shift and so are the residual sums feeding later LayerNorms, so dL/dbias == 0 exactly."""
    p = vit_ref.init_params(TINY, seed=3, randomize=True)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 32, 32, 3))
    _, _, g = torch_ref.loss_and_grads(p, x, rng.integers(0, 10, 3), TINY, 0.1, dtype=torch.float64)
    for l in range(TINY.num_layers):
        assert np.abs(g[f"MixerBlock_{l}/FFBlock_0/Dense_1/bias"]).max() < 1e-12 * np.abs(g[f"MixerBlock_{l}/FFBlock_0/Dense_0/bias"]).max()


def test_bf16_mode_keeps_stream_in_bf16():
    p = vit_ref.init_params(TINY, seed=3, randomize=True)
    x = np.random.default_rng(2).standard_normal((2, 32, 32, 3)).astype(np.float32)
    logits, tokens = vit_ref.mixer_forward(p, x, TINY, mode="bf16", return_tokens=True)
    assert np.array_equal(tokens, vit_ref.bf16_round(tokens)) and np.array_equal(logits, vit_ref.bf16_round(logits))
    ref = vit_ref.forward(p, x, TINY, mode="f32")
    assert 0 < np.abs(logits - ref).max() < 0.1


def test_engine_layout_padding_and_tree():
    import savit_amd  # noqa: F401
    from savit_amd.config import get_config, train_flops_per_image
    from savit_amd.mixer_engine import MixerLayout

    cfg = get_config("mixer_b_patch16")
    lay = MixerLayout(cfg)
    assert (lay.Lp, lay.Fp) == (256, 128)
    flat = torch.zeros(lay.total)
    tree = lay.flax_tree(flat)["params"]
    assert {"params/" + k: tuple(v.shape) for k, v in torch_ref.leaves(tree)} == vit_ref.param_shapes(vit_ref.get_cfg("mixer_b_patch16"))
    # views alias the flat buffer, the padded kernels as strided corners
    v = tree["MixerBlock_3"]["FFBlock_0"]["Dense_0"]["kernel"]
    assert tuple(v.shape) == (196, 98) and v.stride() == (128, 1)
    v.fill_(1.0)
    o, shape = lay.off["l3.tW1"]
    assert shape == (256, 128) and float(flat.sum()) == 196 * 98 and float(flat[o:o + 256 * 128].sum()) == 196 * 98
    # layer slices are contiguous and equally strided (gradient buckets, batched weight casts)
    assert all(b - a == lay.layer_stride for a, b in zip(lay.layer_start, lay.layer_start[1:]))
    assert abs(train_flops_per_image(cfg) - vit_ref.train_flops_per_image(vit_ref.get_cfg("mixer_b_patch16"))) < 1.0


def test_ddp_buckets_cover_the_mixer_layout():
    import savit_amd  # noqa: F401
    from savit_amd.config import get_config
    from savit_amd.ddp import plan_buckets_for
    from savit_amd.mixer_engine import MixerLayout

    lay = MixerLayout(get_config("mixer_s_patch16"))
    buckets = plan_buckets_for(lay, 4 << 20)
    assert buckets[-1][0] == 0 and buckets[0][1] == lay.total and buckets[-1][2] == "Wpe.wgrad"
    assert sum(e - s for s, e, _ in buckets) == lay.total
