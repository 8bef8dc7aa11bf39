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


if __name__ == "__main__":
    make(TINY_VIT, "tiny_vit.npz", 1234)
    make(TINY_CAIT, "tiny_cait.npz", 4321)
    if not os.path.exists(os.path.join(HERE, "tiny_mixer.npz")):  # added later: the two fixtures above are never rewritten
        make(TINY_MIXER, "tiny_mixer.npz", 2468)
    if not os.path.exists(os.path.join(HERE, "tiny_tnt.npz")):
        make(TINY_TNT, "tiny_tnt.npz", 1357)
