"""Is the loss blow-up of tnt_b_patch16 at lr 1e-3 (round 1: 10.37 -> 28.17 in 6 AdamW steps on a fixed batch, gpurun_out/tnt/all.log)
an optimiser instability or a gradient defect?  (dev tool, GPU + host cores)

Runs the SAME experiment twice from identical parameters, images and labels: the HIP engine, and the fp32 torch-CPU restatement of the
reference (oracle/torch_ref.py) driven by the restated optax chain (clip 1.0 -> adam -> additive weight decay -> scale(-lr)), and
prints both loss sequences for lr 1e-3 (no warm-up) and for the reference's scale (5e-4 * batch / 512 with warm-up).
Usage: python tools/tnt_descent_probe.py [batch] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import savit_amd  # noqa: F401
from oracle import torch_ref, vit_ref
from savit_amd.config import get_config
from savit_amd.tnt_engine import TNTEngine


def oracle_run(params0, images, labels, cfg, lrs, wd=1e-4, clip=1.0):
    torch.set_num_threads(torch_ref.host_cores())
    p = torch_ref.to_torch(params0["params"], torch.float32, requires_grad=True)
    leaves = [t for _, t in torch_ref.leaves(p)]
    m = [torch.zeros_like(t) for t in leaves]
    v = [torch.zeros_like(t) for t in leaves]
    x, y = torch.as_tensor(images), torch.as_tensor(labels)
    losses = []
    for step, lr in enumerate(lrs, 1):
        loss = torch_ref.loss_from_logits(torch_ref.forward(p, x, cfg), y)
        losses.append(float(loss))
        g = torch.autograd.grad(loss, leaves, allow_unused=True)
        g = [torch.zeros_like(t) if gi is None else gi for gi, t in zip(g, leaves)]
        norm = float(torch.sqrt(sum((gi.double() ** 2).sum() for gi in g)))
        cs = min(1.0, clip / max(norm, 1e-30))
        with torch.no_grad():
            for t, gi, mi, vi in zip(leaves, g, m, v):
                gi = gi * cs
                mi.mul_(0.9).add_(gi, alpha=0.1)
                vi.mul_(0.999).addcmul_(gi, gi, value=0.001)
                u = (mi / (1 - 0.9 ** step)) / ((vi / (1 - 0.999 ** step)).sqrt() + 1e-8) + wd * t
                t.sub_(lr * u)
    losses.append(float(torch_ref.loss_from_logits(torch_ref.forward(p, x, cfg), y)))
    return losses


def engine_run(params0, images, labels, mc, lrs, wd=1e-4, clip=1.0):
    B = images.shape[0]
    eng = TNTEngine(mc, B)
    eng.load_params(params0)
    x, y = torch.as_tensor(images).cuda(), torch.as_tensor(labels).cuda()
    losses = []
    for lr in lrs:
        eng.forward(x)
        losses.append(float(eng.loss_backward(y, label_smoothing=0.1)))
        eng.optimizer_step(lr=lr, weight_decay=wd, max_norm=clip)
    eng.forward(x)
    losses.append(float(eng.loss_backward(y, label_smoothing=0.1)))
    return losses


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    mc, oc = get_config("tnt_b_patch16"), vit_ref.get_cfg("tnt_b_patch16")
    # the round-1 test's setting: reference initialisers (engine.init_params(42)) + a 0.02-scale stand-in head
    eng0 = TNTEngine(mc, B)
    eng0.init_params(42)
    g = torch.Generator().manual_seed(7)
    eng0.layout.view(eng0.params, "Wh").copy_(torch.randn(mc.embed_dim, mc.num_classes, generator=g) * 0.02)
    params0 = {"params": {k: v for k, v in _to_numpy(eng0.param_tree()["params"]).items()}}
    del eng0
    gd = torch.Generator().manual_seed(1)
    images = vit_ref.bf16_round(torch.randn(B, 224, 224, 3, generator=gd).numpy())
    labels = torch.randint(0, 1000, (B,), generator=gd).numpy()
    for name, lrs in (("lr 1e-3, no warm-up (round-1 test)", [1e-3] * steps),
                      ("reference scale: 5e-4*B/512 peak, linear warm-up", [5e-4 * B / 512 * min(1.0, (k + 1) / steps) for k in range(steps)])):
        le = engine_run(params0, images, labels, mc, lrs)
        lo = oracle_run(params0, images, labels, oc, lrs)
        print(name)
        print("  engine (HIP, bf16):   ", [round(v, 3) for v in le])
        print("  oracle (torch, fp32): ", [round(v, 3) for v in lo], flush=True)


def _to_numpy(tree):
    return {k: (_to_numpy(v) if isinstance(v, dict) else v.detach().cpu().numpy().copy()) for k, v in tree.items()}


if __name__ == "__main__":
    main()
