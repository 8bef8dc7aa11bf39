"""fp32 arithmetic mode (BASELINE config 1: ViT-Tiny/16, fp32; create_model's reference default dtype=float32,
models/create_model.py:6-8): kernels and the end-to-end forward + loss against the fp32 / fp64 oracle.  Bars: kernels vs fp64 on the
same fp32 inputs <= 2e-6 rel-L2 (fp32 summation order); logits of the 12-layer model vs the fp32 oracle <= 2e-5."""
import math

import numpy as np
import pytest
import torch

from oracle import vit_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import savit_amd  # noqa: F401
    from savit_amd import lib

    return lib.load()


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def dev(x):
    return torch.as_tensor(np.asarray(x, np.float32)).cuda()


def st():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("M,N,K,gelu,resid,alpha_cols", [(64, 64, 32, 0, 0, 0), (197 * 3, 576, 192, 0, 0, 192), (100, 768, 192, 1, 0, 0),
                                                         (333, 192, 768, 0, 1, 0), (8, 1000, 192, 0, 0, 0), (1576, 192, 768, 0, 0, 0)])
def test_gemm_f32(L, M, N, K, gelu, resid, alpha_cols):
    rng = np.random.default_rng(M + N + K)
    A, W = rng.standard_normal((M, K)).astype(np.float32), (rng.standard_normal((K, N)) / math.sqrt(K)).astype(np.float32)
    bias, aux = rng.standard_normal(N).astype(np.float32), rng.standard_normal((M, N)).astype(np.float32)
    tA, tW, tb, tx = dev(A), dev(W), dev(bias), dev(aux)
    C = torch.empty(M, N, device="cuda")
    assert L.savit_gemm_f32(tA.data_ptr(), tW.data_ptr(), C.data_ptr(), tb.data_ptr(), tx.data_ptr() if resid else None, M, N, K, K, N, N, N,
                            0.125, alpha_cols, gelu, st()) == 0
    ref = A.astype(np.float64) @ W.astype(np.float64)
    ref[:, :alpha_cols] *= 0.125
    ref += bias
    if gelu:
        ref = 0.5 * ref * (1.0 + np.tanh(math.sqrt(2.0 / math.pi) * (ref + 0.044715 * ref ** 3)))
    if resid:
        ref += aux
    r = rel(C.cpu().numpy(), ref)
    assert r < 2e-6, r


def test_layernorm_and_attention_f32(L):
    rng = np.random.default_rng(3)
    rows, d = 197 * 2, 192
    x = (rng.standard_normal((rows, d)) * 3 + 0.5).astype(np.float32)
    g, b = rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)
    y = torch.empty(rows, d, device="cuda")
    tx, tg, tb = dev(x), dev(g), dev(b)
    assert L.savit_layernorm_fwd_f32(tx.data_ptr(), tg.data_ptr(), tb.data_ptr(), y.data_ptr(), rows, d, d, d, 1e-6, st()) == 0
    pol = vit_ref.Policy("f64")
    assert rel(y.cpu().numpy(), vit_ref.layer_norm(pol, x.astype(np.float64), g, b)) < 2e-6
    for B, N, H, hd in ((2, 197, 3, 64), (3, 50, 4, 48), (1, 256, 2, 64)):
        dm = H * hd
        qkv = rng.standard_normal((B * N, 3 * dm)).astype(np.float32) * 0.6
        o = torch.empty(B * N, dm, device="cuda")
        tq = dev(qkv)
        assert L.savit_attention_fwd_f32(tq.data_ptr(), o.data_ptr(), B, N, H, hd, 3 * dm, st()) == 0
        q, k, v = (qkv[:, i * dm:(i + 1) * dm].reshape(B, N, H, hd).astype(np.float64) for i in range(3))
        s = np.einsum("bqhd,bkhd->bhqk", q, k)
        p = np.exp(s - s.max(-1, keepdims=True))
        p /= p.sum(-1, keepdims=True)
        ref = np.einsum("bhqk,bkhd->bqhd", p, v).reshape(B * N, dm)
        assert rel(o.cpu().numpy(), ref) < 2e-6, (B, N, H, hd)


def test_patchify_and_tokens_f32(L):
    rng = np.random.default_rng(5)
    B, S, P, d = 2, 32, 8, 64
    n = (S // P) ** 2
    img = rng.standard_normal((B, S, S, 3)).astype(np.float32)
    out = torch.empty(B * n, P * P * 3, device="cuda")
    ti = dev(img)
    assert L.savit_patchify_f32(ti.data_ptr(), out.data_ptr(), B, S, P, st()) == 0
    assert np.array_equal(out.cpu().numpy().reshape(B, n, -1), vit_ref.patchify(img, P, P))
    tok, cls, pos = (rng.standard_normal(s).astype(np.float32) for s in ((B * n, d), (d,), (n + 1, d)))
    x0 = torch.empty(B * (n + 1), d, device="cuda")
    tt, tc, tp = dev(tok), dev(cls), dev(pos)
    assert L.savit_assemble_tokens_f32(tt.data_ptr(), tc.data_ptr(), tp.data_ptr(), x0.data_ptr(), B, n + 1, d, st()) == 0
    ref = np.concatenate([np.tile(cls[None, None], (B, 1, 1)), tok.reshape(B, n, d)], axis=1) + pos[None]
    assert np.array_equal(x0.cpu().numpy().reshape(B, n + 1, d), ref.astype(np.float32))


def test_vit_tiny_fp32_forward_and_loss_vs_oracle(L):
    """BASELINE config 1 through the boundary: create_model('vit_ti_patch16') (default dtype float32), batch 8, forward + loss."""
    from savit_amd.model import create_model

    model = create_model("vit_ti_patch16")
    assert model.dtype == torch.float32
    oc = vit_ref.get_cfg("vit_ti_patch16")
    rng = np.random.default_rng(8)
    params = vit_ref.init_params(oc, seed=3, randomize=True)
    images = rng.standard_normal((8, 224, 224, 3)).astype(np.float32)
    labels = rng.integers(0, 1000, 8)
    logits = model.apply(params, torch.as_tensor(images).cuda(), is_training=False)
    assert logits.dtype == torch.float32 and tuple(logits.shape) == (8, 1000)
    other = model.apply(vit_ref.init_params(oc, seed=4, randomize=True), torch.as_tensor(images).cuda(), is_training=False)
    assert not torch.equal(other, logits)  # apply() returns its own tensor, not a view of the engine's reused logits buffer
    logits2 = model.apply(params, torch.as_tensor(images).cuda(), is_training=True)
    assert torch.equal(logits2, logits)  # functional in the parameters; is_training selects nothing (all dropout rates are 0)
    ref32 = vit_ref.forward(params, images, oc, mode="f32")
    ref64 = vit_ref.forward(params, images, oc, mode="f64")
    r32, r64, r_o = rel(logits.cpu().numpy(), ref32), rel(logits.cpu().numpy(), ref64), rel(ref32, ref64)
    print(f"[fp32 vit_ti] logits rel-L2: engine vs fp32 oracle {r32:.2e}, engine vs fp64 oracle {r64:.2e} (fp32 oracle vs fp64 {r_o:.2e})")
    assert r32 < 2e-5 and r64 < 2e-5
    eng = model.engine(8)
    loss = float(eng.loss_fn(torch.as_tensor(labels).cuda(), 0.1))
    assert abs(loss - vit_ref.loss_fn(ref64, labels, 0.1)) < 2e-5 * max(1.0, abs(loss))
    # the reference initialisers (zero head): logits == 0, loss == ln 1000 (SURVEY 8c i) - in fp32 too
    model2 = create_model("vit_ti_patch16", dtype=torch.float32)
    out, p2 = model2.init_with_output(0, torch.ones(2, 224, 224, 3, device="cuda"), is_training=True)
    assert float(out.abs().max()) == 0.0
    assert abs(float(model2.engine(2).loss_fn(torch.tensor([1, 2], device="cuda"), 0.1)) - math.log(1000.0)) < 1e-5
    assert sum(v.numel() for v in _leaves(p2["params"])) == 5_708_008
    with pytest.raises(NotImplementedError, match="bf16"):
        model2.engine(2).loss_backward(torch.tensor([1, 2], device="cuda"))


def test_other_families_state_the_bf16_restriction(L):
    from savit_amd.model import create_model

    for name in ("cait_xxs_24", "mixer_s_patch32", "tnt_s_patch16"):
        with pytest.raises(NotImplementedError, match="dtype=torch.bfloat16"):
            create_model(name)  # the reference default, float32


def _leaves(t):
    for v in t.values():
        if isinstance(v, dict):
            yield from _leaves(v)
        else:
            yield v
