"""Second, independent CPU oracle: a PyTorch-CPU composition of the same path.  TEST
INFRASTRUCTURE ONLY (same rules as oracle/vit_ref.py; "parity unpinned" applies equally).

Purpose: (1) guard against "the restatement copies its own bug" - this file shares no code
with vit_ref.py and is built from torch.nn.functional primitives (F.layer_norm(eps=1e-6),
F.gelu(approximate='tanh'), F.softmax, F.linear with TRANSPOSED Flax kernels,
F.unfold-free patchify); (2) provide gradients of every parameter through torch autograd,
which is what the HIP backward kernels are checked against; (3) serve as the timed
"CPU restatement (JAX absent)" baseline of SURVEY 8d / BASELINE.md section 3.

Citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F


def to_torch(tree, dtype=torch.float32, requires_grad=False):
    if isinstance(tree, dict):
        return {k: to_torch(v, dtype, requires_grad) for k, v in tree.items()}
    t = torch.as_tensor(np.asarray(tree)).to(dtype).clone()
    t.requires_grad_(requires_grad)
    return t


def leaves(tree, prefix=""):
    for k, v in tree.items():
        key = f"{prefix}/{k}" if prefix else k
        if isinstance(v, dict):
            yield from leaves(v, key)
        else:
            yield key, v


def _ln(x, p):  # flax nn.LayerNorm default epsilon=1e-6 (vit.py:19)
    return F.layer_norm(x, (x.shape[-1],), p["scale"], p["bias"], eps=1e-6)


def _attn(p, xq, xkv, H, talking=False):
    """attention.py:21-67 via F.linear on transposed kernels."""
    B, Nq, d = xq.shape
    Nk = xkv.shape[1]
    hd = d // H
    q = F.linear(xq, p["queries"]["kernel"].reshape(d, d).t()).view(B, Nq, H, hd).permute(0, 2, 1, 3)
    k = F.linear(xkv, p["keys"]["kernel"].reshape(d, d).t()).view(B, Nk, H, hd).permute(0, 2, 1, 3)
    v = F.linear(xkv, p["values"]["kernel"].reshape(d, d).t()).view(B, Nk, H, hd).permute(0, 2, 1, 3)
    s = torch.matmul(q / math.sqrt(hd), k.transpose(-1, -2))  # [B,H,Nq,Nk]
    if talking:  # talking_heads.py:13  einsum('h i, b h ... -> b i ...')
        s = torch.einsum("hi,bhqk->biqk", p["TalkingHeadsBlock_0"]["talking_heads_transform"], s)
    w = F.softmax(s, dim=-1)
    if talking:
        w = torch.einsum("hi,bhqk->biqk", p["TalkingHeadsBlock_1"]["talking_heads_transform"], w)
    o = torch.matmul(w, v).permute(0, 2, 1, 3).reshape(B, Nq, d)
    return F.linear(o, p["DenseGeneral_0"]["kernel"].reshape(d, -1).t())


def _ff(p, x):  # ff.py:16-34
    h = F.linear(x, p["Dense_0"]["kernel"].t(), p["Dense_0"]["bias"])
    h = F.gelu(h, approximate="tanh")
    return F.linear(h, p["Dense_1"]["kernel"].t(), p["Dense_1"]["bias"])


def _patchify(images, P):  # patch_embed.py:19-22, c fastest
    B, S, _, C = images.shape
    g = S // P
    x = images.view(B, g, P, g, P, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, g * g, P * P * C)


def _sd(x, mask, rate, is_training):  # stochastic_depth.py:11-28
    if not is_training or rate == 0.0 or mask is None:
        return x
    return x / (1.0 - rate) * mask.view(-1, *([1] * (x.dim() - 1))).to(x.dtype)


def vit_forward(p, images, cfg, taps: Optional[dict] = None):
    """vit.py:73-99.  p = the inner 'params' dict of torch tensors.  taps (optional dict)
    receives intermediate tensors by name for per-kernel checks."""
    x = F.linear(_patchify(images, cfg.patch), p["PatchEmbedBlock_0"]["Dense_0"]["kernel"].t())
    x = torch.cat([p["cls"].expand(x.shape[0], -1, -1), x], dim=1)
    enc = p["Encoder_0"]
    x = x + enc["AddAbsPosEmbed_0"]["pos_embed"]
    if taps is not None:
        taps["x0"] = x
    for l in range(cfg.num_layers):
        b = enc[f"EncoderBlock_{l}"]
        x = x + _attn(b["SelfAttentionBlock_0"], _ln(x, b["LayerNorm_0"]), _ln(x, b["LayerNorm_0"]), cfg.num_heads)
        x = x + _ff(b["FFBlock_0"], _ln(x, b["LayerNorm_1"]))
        if taps is not None:
            taps[f"x{l + 1}"] = x
    z = _ln(x, enc["LayerNorm_0"])
    return F.linear(z[:, 0], p["Dense_0"]["kernel"].t(), p["Dense_0"]["bias"])


def cait_forward(p, images, cfg, is_training=False, keep_masks=None):
    """cait.py:140-183."""
    x = F.linear(_patchify(images, cfg.patch), p["PatchEmbedBlock_0"]["Dense_0"]["kernel"].t())
    enc = p["Encoder_0"]
    x = x + enc["AddAbsPosEmbed_0"]["pos_embed"]
    r = cfg.stoch_depth_rate
    for l in range(cfg.num_layers):
        b = enc[f"EncoderBlock_{l}"]
        mk = None if keep_masks is None else keep_masks[l]
        h = _ln(x, b["LayerNorm_0"])
        a = _attn(b["SelfAttentionBlock_0"], h, h, cfg.num_heads, talking=True) * b["LayerScaleBlock_0"]["layerscale"]
        x = x + _sd(a, None if mk is None else mk[0], r, is_training)
        f = _ff(b["FFBlock_0"], _ln(x, b["LayerNorm_1"])) * b["LayerScaleBlock_1"]["layerscale"]
        x = x + _sd(f, None if mk is None else mk[1], r, is_training)
    cls = p["cls"].expand(x.shape[0], -1, -1)
    for l in range(cfg.num_layers_token_only):
        b = p[f"CAEncoderBlock_{l}"]
        mk = None if keep_masks is None else keep_masks[cfg.num_layers + l]
        h = _ln(torch.cat([cls, x], dim=1), b["LayerNorm_0"])
        a = _attn(b["ClassSelfAttentionBlock_0"], h[:, 0:1], h, cfg.num_heads) * b["LayerScaleBlock_0"]["layerscale"]
        cls = cls + _sd(a, None if mk is None else mk[0], r, is_training)
        f = _ff(b["FFBlock_0"], _ln(cls, b["LayerNorm_1"])) * b["LayerScaleBlock_1"]["layerscale"]
        cls = cls + _sd(f, None if mk is None else mk[1], r, is_training)
    z = _ln(torch.cat([cls, x], dim=1), p["LayerNorm_0"])
    return F.linear(z[:, 0], p["Dense_0"]["kernel"].t(), p["Dense_0"]["bias"])


def mixer_forward(p, images, cfg, taps: Optional[dict] = None):
    """mlp_mixer.py:44-64 (MixerBlock :17-31): token mixing = FFBlock on the transposed activation."""
    pe = p["PatchEmbedBlock_0"]["Dense_0"]
    x = F.linear(_patchify(images, cfg.patch), pe["kernel"].t(), pe["bias"])
    if taps is not None:
        taps["x0"] = x
    for l in range(cfg.num_layers):
        b = p[f"MixerBlock_{l}"]
        x = x + _ff(b["FFBlock_0"], _ln(x, b["LayerNorm_0"]).transpose(1, 2)).transpose(1, 2)
        x = x + _ff(b["FFBlock_1"], _ln(x, b["LayerNorm_1"]))
        if taps is not None:
            taps[f"x{l + 1}"] = x
    z = _ln(x, p["LayerNorm_0"]).mean(dim=1)
    return F.linear(z, p["Dense_0"]["kernel"].t(), p["Dense_0"]["bias"])


def tnt_forward(p, images, cfg, taps: Optional[dict] = None):
    """tnt.py:150-193 (EncoderBlock :66-93, Inner2OuterBlock :40-51, PixelEmbedBlock :17-33)."""
    B, S, _, C = images.shape
    P, t = cfg.patch, cfg.transformed_patch
    g, s = S // P, P // t
    px = images.view(B, g, s, t, g, s, t, C).permute(0, 1, 4, 2, 5, 7, 3, 6).reshape(B * g * g, s * s, C * t * t)
    pe, pa = p["PixelEmbedBlock_0"]["Dense_0"], p["PatchEmbedBlock_0"]["Dense_0"]
    pixels = F.linear(px, pe["kernel"].t(), pe["bias"]) + p["AddAbsPosEmbed_0"]["pos_embed"]
    patches = F.linear(_patchify(images, P), pa["kernel"].t(), pa["bias"])
    patches = torch.cat([p["cls"].expand(B, -1, -1), patches], dim=1) + p["AddAbsPosEmbed_1"]["pos_embed"]
    for l in range(cfg.num_layers):
        b = p["Encoder_0"][f"EncoderBlock_{l}"]
        h = _ln(pixels, b["LayerNorm_0"])
        pixels = pixels + _attn(b["SelfAttentionBlock_0"], h, h, cfg.inner_num_heads)
        pixels = pixels + _ff(b["FFBlock_0"], _ln(pixels, b["LayerNorm_1"]))
        i2o = b["Inner2OuterBlock_0"]["Dense_0"]
        z = F.linear(pixels.reshape(pixels.shape[0], -1), i2o["kernel"].t(), i2o["bias"]).view(B, -1, patches.shape[-1])
        outer = F.pad(z, (0, 0, 1, 0)) + patches
        h = _ln(outer, b["LayerNorm_2"])
        x = patches + _attn(b["SelfAttentionBlock_1"], h, h, cfg.num_heads)  # tnt.py:86: the skip is patch_inputs
        patches = x + _ff(b["FFBlock_1"], _ln(x, b["LayerNorm_3"]))
        if taps is not None:
            taps[f"x{l + 1}"] = patches
    return F.linear(patches[:, 0], p["Dense_0"]["kernel"].t(), p["Dense_0"]["bias"])


def forward(p, images, cfg, is_training=False, keep_masks=None, taps=None):
    if cfg.kind == "vit":
        return vit_forward(p, images, cfg, taps)
    if cfg.kind == "tnt":
        return tnt_forward(p, images, cfg, taps)
    if cfg.kind == "mixer":
        return mixer_forward(p, images, cfg, taps)
    return cait_forward(p, images, cfg, is_training, keep_masks)


def loss_from_logits(logits, labels, label_smoothing=0.1):
    """train.py:83-90: one_hot -> smooth_labels -> softmax_cross_entropy -> mean."""
    K = logits.shape[-1]
    y = F.one_hot(labels, K).to(logits.dtype) * (1.0 - label_smoothing) + label_smoothing / K
    return -(y * F.log_softmax(logits.float() if logits.dtype != torch.float64 else logits, dim=-1)).sum(-1).mean()


def loss_and_grads(params_np: dict, images_np, labels_np, cfg, label_smoothing=0.1, dtype=torch.float32,
                   is_training=False, keep_masks=None, taps=None):
    """Returns (loss float, logits ndarray, {flat-name: grad ndarray}) by autograd."""
    inner = params_np["params"] if "params" in params_np else params_np
    p = to_torch(inner, dtype, requires_grad=True)
    images = torch.as_tensor(np.asarray(images_np)).to(dtype)
    labels = torch.as_tensor(np.asarray(labels_np)).long()
    km = None if keep_masks is None else torch.as_tensor(np.asarray(keep_masks)).to(dtype)
    logits = forward(p, images, cfg, is_training, km, taps)
    loss = loss_from_logits(logits, labels, label_smoothing)
    names, tensors = zip(*leaves(p))
    grads = torch.autograd.grad(loss, tensors, allow_unused=True)
    gd = {n: (np.zeros(tuple(t.shape), np.float64 if dtype == torch.float64 else np.float32) if g is None else g.detach().numpy())
          for n, t, g in zip(names, tensors, grads)}
    return float(loss.detach()), logits.detach().numpy(), gd


def host_cores(cap: int = 16) -> int:
    """Cores this process may really use: the scheduler affinity, capped (a 1-GPU box exposes 256 logical
    CPUs but grants a 16-core share; oversubscribing torch's thread pool makes the timing meaningless)."""
    import os

    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    return max(1, min(n, cap))


def time_train_step(cfg, batch: int, seconds: float = 12.0, threads: Optional[int] = None, seed: int = 0, max_steps: int = 50,
                    backward: bool = True):
    """CPU baseline leg of bench.py: fp32 forward+loss (+backward) of `cfg` at `batch` on this host's
    cores with torch eager; returns dict(images_per_s, cores, steps, seconds).  Bounded: stops after `seconds`
    (checked after every step, the first included) or max_steps."""
    import time
    from . import vit_ref

    threads = threads or host_cores()
    torch.set_num_threads(threads)
    params = vit_ref.init_params(cfg, seed=seed, randomize=True)
    p = to_torch(params["params"], torch.float32, requires_grad=True)
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(batch, cfg.img_size, cfg.img_size, 3, generator=g)
    labels = torch.randint(0, cfg.num_classes, (batch,), generator=g)
    tensors = [t for _, t in leaves(p)]

    def step():
        if backward:
            loss = loss_from_logits(forward(p, images, cfg), labels)
            torch.autograd.grad(loss, tensors, allow_unused=True)
        else:
            with torch.no_grad():
                loss_from_logits(forward(p, images, cfg), labels)

    t0 = time.perf_counter()
    step()  # first step doubles as warm-up; it is kept only if the budget is already gone
    first = time.perf_counter() - t0
    if first >= seconds:
        return {"images_per_s": batch / first, "cores": threads, "steps": 1, "seconds": first}
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= max_steps:
            break
    return {"images_per_s": n * batch / el, "cores": threads, "steps": n, "seconds": el}
