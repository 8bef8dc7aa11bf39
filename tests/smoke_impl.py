"""__graft_entry__.smoke(): one small train step of the hot path on cuda:0, checked against the CPU oracle."""
import numpy as np
import torch


def run():
    from oracle import torch_ref, vit_ref
    import savit_amd  # noqa: F401
    from savit_amd.config import ModelConfig
    from savit_amd.engine import ViTEngine

    kw = dict(kind="vit", num_layers=2, num_heads=3, embed_dim=192, patch=16, num_classes=1000, img_size=224)
    mc, oc = ModelConfig(**kw), vit_ref.Cfg(**kw)
    rng = np.random.default_rng(0)
    params = vit_ref.init_params(oc, seed=1, randomize=True)
    B = 4
    images = vit_ref.bf16_round(rng.standard_normal((B, 224, 224, 3)).astype(np.float32))
    labels = rng.integers(0, 1000, B)
    eng = ViTEngine(mc, B)
    eng.load_params(params)
    logits = eng.forward(torch.as_tensor(images).cuda()).float().cpu().numpy()
    loss = float(eng.loss_backward(torch.as_tensor(labels).cuda(), 0.1))
    eng.optimizer_step(lr=1e-3, weight_decay=1e-4, max_norm=1.0)
    torch.cuda.synchronize()
    ref = vit_ref.forward(params, images, oc, mode="f32")
    r = float(np.linalg.norm(logits - ref) / np.linalg.norm(ref))
    loss_ref, _, g_ref = torch_ref.loss_and_grads(params, images, labels, oc, 0.1)
    assert np.isfinite(logits).all() and r < 1.3e-2, f"logits rel-L2 {r}"
    assert abs(loss - loss_ref) < 3e-2 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    print(f"smoke: logits rel-L2 vs fp32 oracle {r:.2e}; loss {loss:.4f} (oracle {loss_ref:.4f})")
