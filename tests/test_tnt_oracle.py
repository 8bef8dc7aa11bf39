"""Pins the TNT part of the CPU oracle (oracle/vit_ref.py tnt_forward; /root/reference/models/tnt.py) and the host-side layout of the
HIP engine (head-padded inner attention kernels).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import torch_ref, vit_ref

TINY = vit_ref.Cfg(kind="tnt", num_layers=2, num_heads=2, embed_dim=32, patch=16, num_classes=10, img_size=32, inner_num_heads=2, inner_embed_dim=8)


def _closed_form_count(L, Hi, Ho, di, do, C=1000, patch=16, t=4, img=224):
    """Parameter count from the module definitions (tnt.py): pixel + patch Dense with bias, cls, two position embeddings, per layer
    an inner and an outer pre-LN block (bias-free q/k/v/out, FF with biases, two LayerNorms each) and the Inner2Outer Dense."""
    n, npx = (img // patch) ** 2, (patch // t) ** 2

    def block(w):
        return 2 * (2 * w) + 4 * w * w + (w * 4 * w + 4 * w) + (4 * w * w + w)

    emb = (3 * t * t * di + di) + (patch * patch * 3 * do + do) + do + npx * di + (n + 1) * do
    return emb + L * (block(di) + block(do) + (npx * di * do + do)) + do * C + C


@pytest.mark.parametrize("name,args,published", [("tnt_s_patch16", (12, 4, 10, 40, 640), 65.4e6), ("tnt_b_patch16", (12, 4, 6, 24, 384), 23.8e6)])
def test_param_counts_and_geometry(name, args, published):  # create_model.py:50-63
    cfg = vit_ref.get_cfg(name)
    L, Hi, Ho, di, do = args
    assert (cfg.kind, cfg.num_layers, cfg.inner_num_heads, cfg.num_heads, cfg.inner_embed_dim, cfg.embed_dim) == ("tnt", L, Hi, Ho, di, do)
    assert cfg.n_patches == 196 and cfg.n_pixels == 16 and cfg.seq_len == 197
    total = sum(int(np.prod(s)) for s in vit_ref.param_shapes(cfg).values())
    assert total == _closed_form_count(*args)
    # the TNT paper's model sizes (23.8 M and 65.6 M parameters for these two geometries)
    assert abs(total - published) < 0.01 * published


def test_shapes_match_init_and_zero_head():
    p = vit_ref.init_params(TINY, seed=0)
    assert {k: tuple(v.shape) for k, v in vit_ref.flatten(p).items()} == vit_ref.param_shapes(TINY)
    x = np.random.default_rng(0).standard_normal((2, 32, 32, 3)).astype(np.float32)
    logits = vit_ref.forward(p, x, TINY)
    assert logits.shape == (2, 10) and np.all(logits == 0)  # zero-initialised head kernel and bias (tnt.py:189-192)
    assert np.array_equal(logits, vit_ref.forward(p, x, TINY, is_training=True))


def test_pixelify_order():
    """'b (h p1)(w p2) c -> (b h w) p1 p2 c' then 'n (p1 t1)(p2 t2) c -> n (p1 p2) (c t1 t2)' (tnt.py:21-29), against einops."""
    from einops import rearrange

    img = np.arange(2 * 32 * 32 * 3, dtype=np.float32).reshape(2, 32, 32, 3)
    x = rearrange(img, "b (h p1) (w p2) c -> (b h w) p1 p2 c", p1=16, p2=16)
    x = rearrange(x, "n (p1 t1) (p2 t2) c -> n (p1 p2) (c t1 t2)", t1=4, t2=4)
    assert np.array_equal(vit_ref.pixelify(img, 16, 4), x)


def test_two_restatements_agree_and_gradients_flow():
    p = vit_ref.init_params(TINY, seed=3, randomize=True)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 32, 32, 3)).astype(np.float32)
    a = vit_ref.forward(p, x, TINY, mode="f32")
    b = torch_ref.forward(torch_ref.to_torch(p["params"]), torch.as_tensor(x), TINY).numpy()
    assert np.abs(a - b).max() <= 2e-6 * max(1.0, np.abs(b).max())
    _, _, g = torch_ref.loss_and_grads(p, x, rng.integers(0, 10, 3), TINY, 0.1)
    assert all(np.abs(v).max() > 0 for k, v in g.items()), [k for k, v in g.items() if np.abs(v).max() == 0]


def test_outer_residual_skips_the_inner2outer_sum():
    """tnt.py:86 adds `patch_inputs`, not the Inner2Outer sum: with the outer attention's out-projection zeroed, the Inner2Outer
    parameters of that layer cannot influence the output."""
    p = vit_ref.init_params(TINY, seed=4, randomize=True)
    x = np.random.default_rng(2).standard_normal((2, 32, 32, 3)).astype(np.float32)
    last = p["params"]["Encoder_0"][f"EncoderBlock_{TINY.num_layers - 1}"]
    last["SelfAttentionBlock_1"]["DenseGeneral_0"]["kernel"][...] = 0
    a = vit_ref.forward(p, x, TINY)
    last["Inner2OuterBlock_0"]["Dense_0"]["kernel"][...] *= 3.0
    last["Inner2OuterBlock_0"]["Dense_0"]["bias"][...] += 1.0
    assert np.array_equal(a, vit_ref.forward(p, x, TINY))


def test_engine_layout_padding_and_tree():
    import savit_amd  # noqa: F401
    from savit_amd.config import get_config, train_flops_per_image
    from savit_amd.ddp import plan_buckets_for
    from savit_amd.tnt_engine import HDP, TNTLayout

    for name in ("tnt_b_patch16", "tnt_s_patch16"):
        cfg = get_config(name)
        lay = TNTLayout(cfg)
        flat = torch.zeros(lay.total)
        tree = lay.flax_tree(flat)["params"]
        assert {"params/" + k: tuple(v.shape) for k, v in torch_ref.leaves(tree)} == vit_ref.param_shapes(vit_ref.get_cfg(name))
        hd = cfg.inner_embed_dim // cfg.inner_num_heads
        q = tree["Encoder_0"]["EncoderBlock_5"]["SelfAttentionBlock_0"]["queries"]["kernel"]
        assert tuple(q.shape) == (cfg.inner_embed_dim, cfg.inner_num_heads, hd) and q.stride() == (3 * cfg.inner_num_heads * HDP, HDP, 1)
        q.fill_(1.0)  # only the logical corner of the head-padded storage is touched
        assert float(flat.sum()) == cfg.inner_embed_dim * cfg.inner_num_heads * hd
        assert all(b - a == lay.layer_stride for a, b in zip(lay.layer_start, lay.layer_start[1:]))
        buckets = plan_buckets_for(lay, 4 << 20)
        assert buckets[-1][0] == 0 and buckets[0][1] == lay.total and sum(e - s for s, e, _ in buckets) == lay.total
        assert abs(train_flops_per_image(cfg) - vit_ref.train_flops_per_image(vit_ref.get_cfg(name))) < 1.0
