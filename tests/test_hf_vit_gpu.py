"""The HIP engines against an independent implementation of the architecture: HuggingFace transformers' ViT (CPU, fp32) with the SAME
weights, at BASELINE config 1's model (ViT-Ti/16, 12 layers, 224^2, batch 8).  Complements tests/test_oracle_hf_pin.py (which pins
the CPU oracle against the HF model in fp64): here no file of this repository sits on the reference side of the comparison.

  * fp32 engine (create_model's default dtype, models/create_model.py:6-8): logits <= 2e-5 rel-L2 of the HF fp32 forward;
  * bf16 engine (the training path): within the bar the bf16 parity growth curve gives for this model (profiles/
    r02_parity_growth_vit_ti.log: logits 8.6e-3 from the fp32 oracle, the bf16-emulating restatement of the reference itself 8.7e-3;
    measured here against HF: 9.9e-3; bar 1.3e-2 = 1.5 x), and the loss of the label-smoothed cross-entropy within 2e-3."""
import math

import numpy as np
import pytest
import torch

from oracle import vit_ref

pytestmark = pytest.mark.gpu
transformers = pytest.importorskip("transformers")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import savit_amd  # noqa: F401
    from tests.test_oracle_hf_pin import _hf_model, _load

    oc = vit_ref.get_cfg("vit_ti_patch16")
    params = vit_ref.init_params(oc, seed=21, randomize=True)
    rng = np.random.default_rng(2)
    images = vit_ref.bf16_round(rng.standard_normal((8, 224, 224, 3)).astype(np.float32))
    labels = rng.integers(0, 1000, 8)
    hf = _hf_model(oc)
    _load(hf, vit_ref.flatten(params), oc)
    hf = hf.float()
    with torch.no_grad():
        logits = hf(pixel_values=torch.tensor(images).permute(0, 3, 1, 2).contiguous()).logits.numpy()
    logp = logits - np.log(np.exp(logits - logits.max(-1, keepdims=True)).sum(-1, keepdims=True)) - logits.max(-1, keepdims=True)
    target = np.eye(1000)[labels] * 0.9 + 0.1 / 1000
    loss = float(-(target * logp).sum(-1).mean())
    return params, images, labels, logits, loss


def test_fp32_engine_matches_huggingface_vit(setup):
    from savit_amd.model import create_model

    params, images, labels, hf_logits, hf_loss = setup
    model = create_model("vit_ti_patch16")  # dtype defaults to float32, as in the reference
    logits = model.apply(params, torch.as_tensor(images).cuda(), is_training=False).cpu().numpy()
    r = rel(logits, hf_logits)
    print(f"[hf vit_ti fp32] engine vs HuggingFace ViT logits rel-L2 {r:.2e}")
    assert r < 2e-5, r
    loss = float(model.engine(8).loss_fn(torch.as_tensor(labels).cuda(), 0.1))
    assert abs(loss - hf_loss) < 2e-5 * max(1.0, abs(hf_loss)), (loss, hf_loss)


def test_bf16_engine_matches_huggingface_vit(setup):
    from savit_amd.model import create_model

    params, images, labels, hf_logits, hf_loss = setup
    model = create_model("vit_ti_patch16", dtype=torch.bfloat16)
    logits = model.apply(params, torch.as_tensor(images).cuda(), is_training=False).float().cpu().numpy()
    r = rel(logits, hf_logits)
    print(f"[hf vit_ti bf16] engine vs HuggingFace ViT (fp32) logits rel-L2 {r:.2e}")
    assert np.isfinite(logits).all() and r < 1.3e-2, r
    loss = float(model.engine(8).loss_backward(torch.as_tensor(labels).cuda(), 0.1))  # loss of the forward just run (+ its backward)
    assert math.isfinite(loss) and abs(loss - hf_loss) < 2e-3 * max(1.0, abs(hf_loss)), (loss, hf_loss)
