"""Per-layer growth of the bf16 deviation (dev tool, GPU): residual stream entering layer l, relative L2
  engine vs fp32 oracle | bf16-emulating oracle vs fp32 oracle | engine vs bf16-emulating oracle
for a full-depth model on identical weights / inputs (DESIGN.md section 2 quotes this table).
Usage: python tools/parity_growth.py [model] [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import savit_amd  # noqa: F401
from oracle import vit_ref
from savit_amd.config import get_config
from savit_amd.engine import ViTEngine


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def streams(params, images, cfg, mode):
    """Residual stream entering every layer + logits, following vit_ref.vit_forward."""
    pol = vit_ref.Policy(mode)
    p = params["params"]
    x = vit_ref.patchify(pol.lo(images), cfg.patch, cfg.patch)
    x = vit_ref.dense(pol, x, p["PatchEmbedBlock_0"]["Dense_0"]["kernel"])
    cls = np.tile(pol.hi(p["cls"]), (x.shape[0], 1, 1))
    x = np.concatenate([cls, pol.hi(x)], axis=1)
    enc = p["Encoder_0"]
    x = x + pol.hi(enc["AddAbsPosEmbed_0"]["pos_embed"])
    out = [x]
    for l in range(cfg.num_layers):
        x = vit_ref.vit_encoder_block(pol, enc[f"EncoderBlock_{l}"], x, cfg.num_heads)
        out.append(x)
    z = vit_ref.layer_norm(pol, x, enc["LayerNorm_0"]["scale"], enc["LayerNorm_0"]["bias"])
    return out, vit_ref.dense(pol, z[:, 0], p["Dense_0"]["kernel"], p["Dense_0"]["bias"])


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "vit_ti_patch16"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    mc, oc = get_config(model), vit_ref.get_cfg(model)
    params = vit_ref.init_params(oc, seed=5, randomize=True)
    rng = np.random.default_rng(11)
    images = vit_ref.bf16_round(rng.standard_normal((B, oc.img_size, oc.img_size, 3)).astype(np.float32))
    os.environ["SAVIT_CLS_ONLY_LAST"] = "0"  # this tool reads the whole residual stream: the dense plan writes every row of the last layer
    eng = ViTEngine(mc, B)
    eng.load_params(params)
    logits = eng.forward(torch.as_tensor(images).cuda()).float().cpu().numpy()
    x32, l32 = streams(params, images, oc, "f32")
    xbf, lbf = streams(params, images, oc, "bf16")
    print(f"{model} B={B}: relative L2 of the residual stream entering layer l (last row: logits)")
    print("layer | engine vs fp32 | bf16-emulation vs fp32 | engine vs bf16-emulation")
    for l in range(oc.num_layers + 1):
        xe = eng.x[l].float().cpu().numpy().reshape(x32[l].shape)
        print(f"{l:5d} | {rel(xe, x32[l]):.2e} | {rel(xbf[l], x32[l]):.2e} | {rel(xe, xbf[l]):.2e}")
    print(f"logit | {rel(logits, l32):.2e} | {rel(lbf, l32):.2e} | {rel(logits, lbf):.2e}")


if __name__ == "__main__":
    main()
