"""Generates the committed golden fixtures from the CPU oracle (oracle/vit_ref.py, fp64 master,
stored fp32) - the reference itself cannot be imported here (no jax), so these pin the
ORACLE against regression, not the reference.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch  # noqa: E402

from oracle import torch_ref, vit_ref  # noqa: E402

TINY_VIT = vit_ref.Cfg(kind="vit", num_layers=2, num_heads=2, embed_dim=32, patch=8, num_classes=10, img_size=32)
TINY_CAIT = vit_ref.Cfg(kind="cait", num_layers=2, num_heads=2, embed_dim=32, patch=8, num_classes=10, img_size=32,
                        num_layers_token_only=2, stoch_depth_rate=0.1, layerscale_eps=1e-5)


TINY_MIXER = vit_ref.Cfg(kind="mixer", num_layers=2, num_heads=1, embed_dim=32, patch=8, num_classes=10, img_size=32)
TINY_TNT = vit_ref.Cfg(kind="tnt", num_layers=2, num_heads=2, embed_dim=32, patch=16, num_classes=10, img_size=32, inner_num_heads=2,
                       inner_embed_dim=8)


def make(cfg, name, seed):
    rng = np.random.default_rng(seed)
    params = vit_ref.init_params(cfg, seed=seed, randomize=True)
    B = 3
    images = rng.standard_normal((B, cfg.img_size, cfg.img_size, 3)).astype(np.float32)
    labels = rng.integers(0, cfg.num_classes, size=B)
    logits64 = vit_ref.forward(params, images, cfg, mode="f64")
    logits_bf16 = vit_ref.forward(params, images, cfg, mode="bf16")
    loss64 = vit_ref.loss_fn(logits64, labels, 0.1)
    _, _, grads = torch_ref.loss_and_grads(params, images, labels, cfg, 0.1, dtype=torch.float64)
    out = {"images": images, "labels": labels.astype(np.int64), "logits": logits64.astype(np.float32),
           "logits_bf16": logits_bf16.astype(np.float32), "loss": np.float32(loss64)}
    for k, v in vit_ref.flatten(params).items():
        out["P:" + k] = v
    for k, v in grads.items():
        out["G:params/" + k] = v.astype(np.float32)
    if cfg.kind == "cait":  # a training-mode forward with explicit stochastic-depth masks
        masks = (rng.random((cfg.num_layers + cfg.num_layers_token_only, 2, B)) < 0.7).astype(np.float32)
        out["keep_masks"] = masks
        out["logits_train"] = vit_ref.forward(params, images, cfg, mode="f64", is_training=True,
                                              keep_masks=masks).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "loss", float(loss64), "bytes", os.path.getsize(os.path.join(HERE, name)))


# ---- real-width single-block fixtures (SURVEY 8c "Golden vectors"): one encoder block at the widths / sequence lengths of the
# BASELINE configs.  Parameters and images are NOT stored (8 - 15 M values each): they are regenerated from the seed with numpy's
# PCG64 stream (stable across numpy versions by policy) and pinned by per-tensor checksums; stored are the labels, the fp64 and
# bf16-emulated logits, the loss, and for every parameter gradient its norm plus 2048 sampled entries.
BLOCKS = {
    # DeiT-B/16 width: d 768, 12 heads, N 197 (BASELINE config 3)
    "block_d768_n197": (dict(kind="vit", num_layers=1, num_heads=12, embed_dim=768, patch=16, num_classes=1000, img_size=224), 9001),
    # ViT-L/16 at 384^2: d 1024, 16 heads, N 577 (config 5: the online-softmax attention kernels)
    "block_d1024_n577": (dict(kind="vit", num_layers=1, num_heads=16, embed_dim=1024, patch=16, num_classes=1000, img_size=384), 9002),
    # CaiT-S width: d 384, 8 heads of 48, N 196, talking heads + LayerScale + one class-attention block (config 4)
    "block_cait_d384_n196": (dict(kind="cait", num_layers=1, num_heads=8, embed_dim=384, patch=16, num_classes=1000, img_size=224,
                                  num_layers_token_only=1, stoch_depth_rate=0.1, layerscale_eps=1e-6), 9003),
    # DeiT-S width: d 384, 6 heads (config 2) and ViT-Ti width: d 192, 3 heads (config 1)
    "block_d384_n197": (dict(kind="vit", num_layers=1, num_heads=6, embed_dim=384, patch=16, num_classes=1000, img_size=224), 9004),
    "block_d192_n197": (dict(kind="vit", num_layers=1, num_heads=3, embed_dim=192, patch=16, num_classes=1000, img_size=224), 9005),
    # Round 6: TWO-layer models at the real widths.  Since round 5 the LAST encoder layer runs on the cls rows
    # (engine.cls_only_last / cls_fwd), so in the one-layer fixtures above no token goes through the dense block by default; here layer 0
    # is the dense block (dense attention incl. the N = 577 general kernels, proj / fc1-GELU / fc2-residual, GELU', dense LayerNorm
    # backward) and layer 1 the cls-row path - every row goes through a block as in vit.py:17-32.
    "block2_d768_n197": (dict(kind="vit", num_layers=2, num_heads=12, embed_dim=768, patch=16, num_classes=1000, img_size=224), 9021),
    "block2_d1024_n577": (dict(kind="vit", num_layers=2, num_heads=16, embed_dim=1024, patch=16, num_classes=1000, img_size=384), 9022),
    "block2_d384_n197": (dict(kind="vit", num_layers=2, num_heads=6, embed_dim=384, patch=16, num_classes=1000, img_size=224), 9024),
    # small end-to-end models of all four families at sizes the HIP engines accept (the d = 32 tiny_*.npz models above are below the
    # engines' minimum width; they pin the oracle only)
    "e2e_vit_d128": (dict(kind="vit", num_layers=2, num_heads=2, embed_dim=128, patch=8, num_classes=16, img_size=32), 9011),
    "e2e_cait_d128": (dict(kind="cait", num_layers=2, num_heads=2, embed_dim=128, patch=8, num_classes=16, img_size=32,
                           num_layers_token_only=2, stoch_depth_rate=0.1, layerscale_eps=1e-5), 9012),
    "e2e_mixer_d128": (dict(kind="mixer", num_layers=2, num_heads=1, embed_dim=128, patch=8, num_classes=16, img_size=32), 9013),
    "e2e_tnt_d128": (dict(kind="tnt", num_layers=2, num_heads=2, embed_dim=128, patch=16, num_classes=16, img_size=32, inner_num_heads=4,
                          inner_embed_dim=24), 9014),
}
BLOCK_B = 2
N_SAMPLES = 2048


def block_inputs(cfg_kw, seed):
    """(cfg, params, images, labels) of a block fixture, regenerated from its seed."""
    cfg = vit_ref.Cfg(**cfg_kw)
    params = vit_ref.init_params(cfg, seed=seed, randomize=True)
    rng = np.random.default_rng(seed + 1)
    images = vit_ref.bf16_round(rng.standard_normal((BLOCK_B, cfg.img_size, cfg.img_size, 3)).astype(np.float32))
    labels = rng.integers(0, cfg.num_classes, size=BLOCK_B)
    return cfg, params, images, labels


def checksums(params, images):
    out = {}
    for k, v in vit_ref.flatten(params).items():
        v = np.asarray(v, np.float64)
        out["C:" + k] = np.array([v.sum(), (v * v).sum()])
    im = np.asarray(images, np.float64)
    out["C:images"] = np.array([im.sum(), (im * im).sum()])
    return out


def make_block(name):
    import json

    cfg_kw, seed = BLOCKS[name]
    cfg, params, images, labels = block_inputs(cfg_kw, seed)
    logits64 = vit_ref.forward(params, images, cfg, mode="f64")
    logits_bf16 = vit_ref.forward(params, images, cfg, mode="bf16")
    logits_engine = vit_ref.forward(params, images, cfg, mode="engine")
    loss64 = vit_ref.loss_fn(logits64, labels, 0.1)
    _, _, grads = torch_ref.loss_and_grads(params, images, labels, cfg, 0.1, dtype=torch.float64)
    out = {"cfg": np.array(json.dumps(cfg_kw)), "seed": np.int64(seed), "labels": labels.astype(np.int64), "logits": logits64.astype(np.float64),
           "logits_bf16": logits_bf16.astype(np.float32), "logits_engine": logits_engine.astype(np.float32), "loss": np.float64(loss64)}
    out.update(checksums(params, images))
    srng = np.random.default_rng(seed + 2)
    for k, g in grads.items():
        g = np.asarray(g, np.float64).ravel()
        idx = np.sort(srng.choice(g.size, size=min(N_SAMPLES, g.size), replace=False))
        out["GN:" + k] = np.float64(np.linalg.norm(g))
        out["GI:" + k] = idx.astype(np.int64)
        out["GV:" + k] = g[idx].astype(np.float32)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "loss", float(loss64), "bytes", os.path.getsize(path))


def add_engine_logits(name):
    """Round 3: a third logits array - the oracle with the ENGINE's rounding points (vit_ref.Policy('engine')) - added to a block
    fixture written in round 2.  Every array already in the file is kept byte for byte."""
    path = os.path.join(HERE, name + ".npz")
    old = dict(np.load(path))
    if "logits_engine" in old:
        return
    cfg_kw, seed = BLOCKS[name]
    cfg, params, images, labels = block_inputs(cfg_kw, seed)
    for k, v in checksums(params, images).items():
        np.testing.assert_allclose(v, old[k], rtol=1e-12, atol=1e-12, err_msg=k)
    assert np.allclose(vit_ref.forward(params, images, cfg, mode="bf16"), old["logits_bf16"], rtol=0, atol=0)
    old["logits_engine"] = vit_ref.forward(params, images, cfg, mode="engine").astype(np.float32)
    np.savez_compressed(path, **old)
    print(name, "+ logits_engine", "bytes", os.path.getsize(path))


if __name__ == "__main__":
    for _name in BLOCKS:
        if not os.path.exists(os.path.join(HERE, _name + ".npz")):
            make_block(_name)
        add_engine_logits(_name)
    make(TINY_VIT, "tiny_vit.npz", 1234)
    make(TINY_CAIT, "tiny_cait.npz", 4321)
    if not os.path.exists(os.path.join(HERE, "tiny_mixer.npz")):  # added later: the two fixtures above are never rewritten
        make(TINY_MIXER, "tiny_mixer.npz", 2468)
    if not os.path.exists(os.path.join(HERE, "tiny_tnt.npz")):
        make(TINY_TNT, "tiny_tnt.npz", 1357)
