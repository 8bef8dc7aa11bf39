"""Cross-validation of the CPU oracle's ViT path against an independent third-party implementation of the same published
architecture: HuggingFace `transformers.ViTForImageClassification` (pre-LN encoder, tanh-GELU MLP, cls token + learned position
embedding, final LayerNorm, linear head - what models/vit.py:61-99 of the reference implements in Flax).

This is NOT the reference (JAX / Flax are not installed here): the oracle stays "unpinned" with respect to NZ99's code.
What it excludes is a shared misreading between oracle/vit_ref.py (NumPy) and oracle/torch_ref.py (torch), which were written by
the same hand: the HF model was not.  Same weights (mapped below), same image -> logits to 1e-10 in fp64, and every parameter
gradient of the label-smoothed loss to 1e-8."""
import numpy as np
import pytest
import torch

from oracle import torch_ref, vit_ref

transformers = pytest.importorskip("transformers")


def _hf_model(cfg):
    from transformers import ViTConfig, ViTForImageClassification

    hc = ViTConfig(hidden_size=cfg.embed_dim, num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads,
                   intermediate_size=4 * cfg.embed_dim, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6, image_size=cfg.img_size,
                   patch_size=cfg.patch, num_labels=cfg.num_classes, qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    return ViTForImageClassification(hc).double().eval()


def _load(model, flat, cfg):
    """Flax-shaped oracle parameters -> the HF module (the reference's q/k/v/out projections and patch embedding carry no bias: zeros)."""
    d, H, P = cfg.embed_dim, cfg.num_heads, cfg.patch
    t = lambda a: torch.tensor(np.asarray(a, np.float64))  # noqa: E731
    sd = model.state_dict()
    new = {k: torch.zeros_like(v) for k, v in sd.items()}
    # patchify (vit_ref.patchify) orders a patch vector (row, column, channel); the conv kernel is [out, channel, row, column]
    new["vit.embeddings.patch_embeddings.projection.weight"] = t(flat["params/PatchEmbedBlock_0/Dense_0/kernel"]).reshape(P, P, 3, d).permute(3, 2, 0, 1)
    new["vit.embeddings.cls_token"] = t(flat["params/cls"])
    new["vit.embeddings.position_embeddings"] = t(flat["params/Encoder_0/AddAbsPosEmbed_0/pos_embed"])
    for l in range(cfg.num_layers):
        o, h = f"params/Encoder_0/EncoderBlock_{l}/", f"vit.layers.{l}."
        new[h + "layernorm_before.weight"], new[h + "layernorm_before.bias"] = t(flat[o + "LayerNorm_0/scale"]), t(flat[o + "LayerNorm_0/bias"])
        new[h + "layernorm_after.weight"], new[h + "layernorm_after.bias"] = t(flat[o + "LayerNorm_1/scale"]), t(flat[o + "LayerNorm_1/bias"])
        for ours, theirs in (("queries", "q_proj"), ("keys", "k_proj"), ("values", "v_proj")):
            new[h + f"attention.{theirs}.weight"] = t(flat[o + f"SelfAttentionBlock_0/{ours}/kernel"]).reshape(d, d).T
        new[h + "attention.o_proj.weight"] = t(flat[o + "SelfAttentionBlock_0/DenseGeneral_0/kernel"]).reshape(d, d).T
        new[h + "mlp.fc1.weight"], new[h + "mlp.fc1.bias"] = t(flat[o + "FFBlock_0/Dense_0/kernel"]).T, t(flat[o + "FFBlock_0/Dense_0/bias"])
        new[h + "mlp.fc2.weight"], new[h + "mlp.fc2.bias"] = t(flat[o + "FFBlock_0/Dense_1/kernel"]).T, t(flat[o + "FFBlock_0/Dense_1/bias"])
    new["vit.layernorm.weight"], new["vit.layernorm.bias"] = t(flat["params/Encoder_0/LayerNorm_0/scale"]), t(flat["params/Encoder_0/LayerNorm_0/bias"])
    new["classifier.weight"], new["classifier.bias"] = t(flat["params/Dense_0/kernel"]).T, t(flat["params/Dense_0/bias"])
    assert set(new) == set(sd) and all(new[k].shape == sd[k].shape for k in sd), [k for k in sd if new[k].shape != sd[k].shape]
    model.load_state_dict({k: v.contiguous() for k, v in new.items()})
    return H


@pytest.mark.parametrize("kw", [dict(num_layers=2, num_heads=2, embed_dim=64, patch=8, num_classes=10, img_size=32),
                                dict(num_layers=3, num_heads=3, embed_dim=96, patch=16, num_classes=7, img_size=48)])
def test_oracle_vit_matches_huggingface_vit(kw):
    cfg = vit_ref.Cfg(kind="vit", **kw)
    params = vit_ref.init_params(cfg, seed=11, randomize=True)
    flat = vit_ref.flatten(params)
    rng = np.random.default_rng(5)
    B = 3
    images = rng.standard_normal((B, cfg.img_size, cfg.img_size, 3))
    labels = rng.integers(0, cfg.num_classes, B)
    model = _hf_model(cfg)
    _load(model, flat, cfg)
    x = torch.tensor(images).permute(0, 3, 1, 2).contiguous()  # NHWC -> NCHW
    hf_logits = model(pixel_values=x).logits
    ours = vit_ref.forward(params, images, cfg, mode="f64")
    rel = float(np.linalg.norm(hf_logits.detach().numpy() - ours) / np.linalg.norm(ours))
    assert rel < 1e-10, rel

    # gradients of the label-smoothed cross-entropy (train.py:83-90): oracle autograd composition vs HF autograd
    smooth = 0.1
    logp = torch.log_softmax(hf_logits, dim=-1)
    onehot = torch.nn.functional.one_hot(torch.tensor(labels), cfg.num_classes).double()
    target = onehot * (1.0 - smooth) + smooth / cfg.num_classes
    loss_hf = -(target * logp).sum(-1).mean()
    loss_hf.backward()
    loss_or, _, grads = torch_ref.loss_and_grads(params, images, labels, cfg, smooth, dtype=torch.float64)
    assert abs(float(loss_hf.detach()) - float(loss_or)) < 1e-10 * max(1.0, abs(float(loss_or)))
    hg = dict(model.named_parameters())
    d, P = cfg.embed_dim, cfg.patch
    pairs = {
        "PatchEmbedBlock_0/Dense_0/kernel": hg["vit.embeddings.patch_embeddings.projection.weight"].grad.permute(2, 3, 1, 0).reshape(P * P * 3, d),
        "cls": hg["vit.embeddings.cls_token"].grad,
        "Encoder_0/AddAbsPosEmbed_0/pos_embed": hg["vit.embeddings.position_embeddings"].grad,
        "Encoder_0/EncoderBlock_0/SelfAttentionBlock_0/queries/kernel": hg["vit.layers.0.attention.q_proj.weight"].grad.T.reshape(d, cfg.num_heads, -1),
        "Encoder_0/EncoderBlock_1/SelfAttentionBlock_0/DenseGeneral_0/kernel": hg["vit.layers.1.attention.o_proj.weight"].grad.T.reshape(cfg.num_heads, -1, d),
        "Encoder_0/EncoderBlock_1/FFBlock_0/Dense_0/kernel": hg["vit.layers.1.mlp.fc1.weight"].grad.T,
        "Encoder_0/EncoderBlock_0/FFBlock_0/Dense_1/bias": hg["vit.layers.0.mlp.fc2.bias"].grad,
        "Encoder_0/EncoderBlock_0/LayerNorm_1/scale": hg["vit.layers.0.layernorm_after.weight"].grad,
        "Encoder_0/LayerNorm_0/bias": hg["vit.layernorm.bias"].grad,
        "Dense_0/kernel": hg["classifier.weight"].grad.T,
    }
    for name, g_hf in pairs.items():
        g_or = np.asarray(grads[name], np.float64)
        r = float(np.linalg.norm(g_hf.numpy().reshape(g_or.shape) - g_or) / max(np.linalg.norm(g_or), 1e-30))
        assert r < 1e-8, (name, r)
