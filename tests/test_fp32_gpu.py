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


@pytest.mark.parametrize("name", ["mixer_s_patch32", "tnt_s_patch16"])
def test_mixer_and_tnt_fp32_default_dtype_vs_oracle(L, name):
    """create_model(name) at the reference's default dtype for the two remaining families (models/mlp_mixer.py:44-64, models/tnt.py:150-193):
    logits within 2e-5 of the fp32 oracle (their fp32 train steps: test_mixer_fp32_train_step_vs_autograd,
    test_tnt_fp32_train_step_vs_autograd)."""
    from savit_amd.model import create_model

    model = create_model(name)
    assert model.dtype == torch.float32
    oc = vit_ref.get_cfg(name)
    rng = np.random.default_rng(51)
    params = vit_ref.init_params(oc, seed=11, randomize=True)
    images = rng.standard_normal((3, 224, 224, 3)).astype(np.float32)
    logits = model.apply(params, torch.as_tensor(images).cuda(), is_training=False)
    assert logits.dtype == torch.float32 and tuple(logits.shape) == (3, 1000)
    ref32 = vit_ref.forward(params, images, oc, mode="f32")
    ref64 = vit_ref.forward(params, images, oc, mode="f64")
    r32, r64 = rel(logits.cpu().numpy(), ref32), rel(logits.cpu().numpy(), ref64)
    print(f"[fp32 {name}] logits rel-L2: engine vs fp32 oracle {r32:.2e}, vs fp64 oracle {r64:.2e} (fp32 oracle vs fp64 {rel(ref32, ref64):.2e})")
    assert r32 < 2e-5 and r64 < 2e-5
    labels = rng.integers(0, 1000, 3)
    loss = float(model.engine(3).loss_fn(torch.as_tensor(labels).cuda(), 0.1))
    assert abs(loss - vit_ref.loss_fn(ref64, labels, 0.1)) < 2e-5 * max(1.0, abs(loss))
    # the reference initialisers through the same boundary (mlp_mixer_test.py / tnt_test.py shapes)
    out, _ = create_model(name).init_with_output(0, torch.ones(2, 224, 224, 3, device="cuda"), is_training=False)
    assert tuple(out.shape) == (2, 1000) and bool(torch.isfinite(out).all())


def _gemm_ex(L, **kw):
    from savit_amd import lib

    g = lib.GemmF32Args()
    for k, v in kw.items():
        setattr(g, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    for k in ("batch", "inner", "rows_per_sample"):
        if not getattr(g, k):
            setattr(g, k, 1)
    if "alpha" not in kw:
        g.alpha = 1.0
    import ctypes

    assert L.savit_gemm_f32_ex(ctypes.byref(g), st()) == 0


def _gelu(u):
    return 0.5 * u * (1.0 + np.tanh(math.sqrt(2.0 / math.pi) * (u + 0.044715 * u ** 3)))


def _gelu_grad(u):
    c = math.sqrt(2.0 / math.pi)
    t = np.tanh(c * (u + 0.044715 * u ** 3))
    return 0.5 * (1 + t) + 0.5 * u * (1 - t * t) * c * (1 + 3 * 0.044715 * u * u)


def test_gemm_f32_ex_forms(L):
    """Every form the fp32 engines use: transposed operands (dgrad / wgrad), accumulation, batched (image, head) products with
    outer / inner strides, GELU with the pre-activation kept, GELU' of a saved pre-activation, LayerScale x stochastic-depth x
    residual, a position table shared by the batches."""
    rng = np.random.default_rng(17)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    # wgrad: dW[K, N] += X[M, K]^T dY[M, N]
    M, K, N = 333, 70, 130
    X, dY, dW0 = f(M, K), f(M, N), f(K, N)
    tX, tdY, tdW = dev(X), dev(dY), dev(dW0)
    _gemm_ex(L, A=tX, W=tdY, C=tdW, M=K, N=N, K=M, lda=K, ldw=N, ldc=N, transA=1, accumulate=1)
    assert rel(tdW.cpu().numpy(), dW0 + X.astype(np.float64).T @ dY) < 2e-6
    # dgrad with GELU': dX[M, K] = (dY[M, N] W[K, N]^T) * gelu'(U)
    W, U = f(K, N) / 8, f(M, K)
    tW, tU, tdX = dev(W), dev(U), torch.empty(M, K, device="cuda")
    _gemm_ex(L, A=tdY, W=tW, C=tdX, U=tU, M=M, N=K, K=N, lda=N, ldw=N, ldc=K, transW=1, act=2)
    assert rel(tdX.cpu().numpy(), (dY.astype(np.float64) @ W.T) * _gelu_grad(U.astype(np.float64))) < 2e-6
    # fc1: a = gelu(u), u = X W + b kept
    b = f(N)
    tb, ta, tu = dev(b), torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    _gemm_ex(L, A=tX, W=tW, C=ta, C2=tu, bias=tb, M=M, N=N, K=K, lda=K, ldw=N, ldc=N, act=1)
    u = X.astype(np.float64) @ W + b
    assert rel(tu.cpu().numpy(), u) < 2e-6 and rel(ta.cpu().numpy(), _gelu(u)) < 2e-6
    # residual + LayerScale + stochastic depth: C = aux + ls[n] * keep[m // rows] * (A W + b), 9 rows per sample
    Ms = 9 * 37
    ls, keep, aux = f(N), (rng.random(37) < 0.6).astype(np.float32) / 0.6, f(Ms, N)
    tls, tk, tx, tc = dev(ls), dev(keep), dev(aux), torch.empty(Ms, N, device="cuda")
    _gemm_ex(L, A=tX, W=tW, C=tc, bias=tb, aux=tx, ldaux=N, colscale=tls, rowscale=tk, rows_per_sample=9, M=Ms, N=N, K=K, lda=K, ldw=N, ldc=N)
    assert rel(tc.cpu().numpy(), aux + ls * np.repeat(keep, 9)[:, None] * (X.astype(np.float64) @ W + b)) < 2e-6
    # a table shared by the images: C[m] = A W + pos[m % 37]
    pos = f(37, N)
    tp = dev(pos)
    _gemm_ex(L, A=tX, W=tW, C=tc, aux=tp, ldaux=N, aux_row_mod=37, M=Ms, N=N, K=K, lda=K, ldw=N, ldc=N)
    assert rel(tc.cpu().numpy(), X.astype(np.float64) @ W + np.tile(pos, (9, 1))) < 2e-6
    # batched scores / PV / dK over (image, head) of a fused qkv buffer [B N, 3 d]
    B, Nq, H, hd = 3, 50, 4, 24
    d = H * hd
    qkv = f(B * Nq, 3 * d)
    tq, ts = dev(qkv), torch.empty(B * H, Nq, Nq, device="cuda")
    _gemm_ex(L, A=tq, W=tq.data_ptr() + 4 * d, C=ts, M=Nq, N=Nq, K=hd, lda=3 * d, ldw=3 * d, ldc=Nq, transW=1, batch=B * H, inner=H,
             sAo=Nq * 3 * d, sAi=hd, sWo=Nq * 3 * d, sWi=hd, sCo=H * Nq * Nq, sCi=Nq * Nq, alpha=0.25, alpha_cols=Nq)
    q, k, v = (qkv[:, i * d:(i + 1) * d].reshape(B, Nq, H, hd).astype(np.float64) for i in range(3))
    sref = 0.25 * np.einsum("bqhd,bkhd->bhqk", q, k)
    assert rel(ts.cpu().numpy().reshape(B, H, Nq, Nq), sref) < 2e-6
    to = torch.empty(B * Nq, d, device="cuda")
    _gemm_ex(L, A=ts, W=tq.data_ptr() + 8 * d, C=to, M=Nq, N=hd, K=Nq, lda=Nq, ldw=3 * d, ldc=d, batch=B * H, inner=H,
             sAo=H * Nq * Nq, sAi=Nq * Nq, sWo=Nq * 3 * d, sWi=hd, sCo=Nq * d, sCi=hd)
    assert rel(to.cpu().numpy(), np.einsum("bhqk,bkhd->bqhd", sref, v).reshape(B * Nq, d)) < 2e-6
    tdk = torch.zeros(B * Nq, 3 * d, device="cuda")
    _gemm_ex(L, A=ts, W=tq, C=tdk.data_ptr() + 4 * d, M=Nq, N=hd, K=Nq, lda=Nq, ldw=3 * d, ldc=3 * d, transA=1, batch=B * H, inner=H,
             sAo=H * Nq * Nq, sAi=Nq * Nq, sWo=Nq * 3 * d, sWi=hd, sCo=Nq * 3 * d, sCi=hd)
    got = tdk.cpu().numpy()
    assert rel(got[:, d:2 * d], np.einsum("bhqk,bqhd->bkhd", sref, q).reshape(B * Nq, d)) < 2e-6
    assert not got[:, :d].any() and not got[:, 2 * d:].any()


def test_softmax_headmix_layernorm_bwd_colsum_xent_f32(L):
    rng = np.random.default_rng(23)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    rows, N = 411, 197
    x, dp = f(rows, N) * 3, f(rows, N)
    tx, ty, tdp = dev(x), torch.empty(rows, N, device="cuda"), dev(dp)
    assert L.savit_softmax_rows_f32(tx.data_ptr(), ty.data_ptr(), rows, N, N, st()) == 0
    x64 = x.astype(np.float64)
    p = np.exp(x64 - x64.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    assert rel(ty.cpu().numpy(), p) < 2e-6
    assert L.savit_softmax_rows_bwd_f32(ty.data_ptr(), tdp.data_ptr(), tdp.data_ptr(), rows, N, N, st()) == 0  # in place, as the engine calls it
    assert rel(tdp.cpu().numpy(), p * (dp - (dp * p).sum(-1, keepdims=True))) < 4e-6
    # talking heads (talking_heads.py:13): y[b, i] = sum_h T[h, i] x[b, h]
    for B, H, E in ((3, 4, 17 * 17), (2, 16, 100), (1, 1, 33)):
        T, s = f(H, H), f(B, H, E)
        tT, ts_, to = dev(T), dev(s), torch.empty(B, H, E, device="cuda")
        assert L.savit_head_mix_f32(tT.data_ptr(), ts_.data_ptr(), to.data_ptr(), B, H, E, st()) == 0
        assert rel(to.cpu().numpy(), np.einsum("hi,bhe->bie", T.astype(np.float64), s)) < 2e-6
    # LayerNorm VJP: contiguous rows with the residual cotangent added in place; strided rows (cls rows of [B, N, d]); a row wider than
    # the register-partials form (d > 1024)
    for rows, d, xs in ((197 * 3, 192, 192), (5, 64, 64 * 17), (37, 1280, 1280)):
        xx, dy, g, add = f(rows, xs) * 2 + 0.3, f(rows, d), f(d), f(rows, xs)
        txx, tdy, tg, tadd = dev(xx), dev(dy), dev(g), dev(add)
        tdg, tdb = torch.full((d,), 1.0, device="cuda"), torch.full((d,), -2.0, device="cuda")  # accumulated into
        assert L.savit_layernorm_bwd_f32(tdy.data_ptr(), txx.data_ptr(), tg.data_ptr(), tadd.data_ptr(), tadd.data_ptr(), tdg.data_ptr(), tdb.data_ptr(),
                                         rows, d, xs, d, 1e-6, st()) == 0
        xr = xx[:, :d].astype(np.float64)
        mu, var = xr.mean(-1, keepdims=True), xr.var(-1, keepdims=True)
        rstd = 1.0 / np.sqrt(var + 1e-6)
        xh = (xr - mu) * rstd
        gy = dy.astype(np.float64) * g
        dx = rstd * (gy - gy.mean(-1, keepdims=True) - xh * (gy * xh).mean(-1, keepdims=True))
        got = tadd.cpu().numpy()
        assert rel(got[:, :d], dx + add[:, :d]) < 4e-6, (rows, d)
        assert np.array_equal(got[:, d:], add[:, d:])  # the other tokens' rows are untouched
        assert rel(tdg.cpu().numpy(), 1.0 + (dy * xh).sum(0)) < 4e-6 and rel(tdb.cpu().numpy(), -2.0 + dy.astype(np.float64).sum(0)) < 4e-6
    M, Nc = 5000, 300
    y = f(M, Nc)
    ty, tout = dev(y), torch.ones(Nc, device="cuda")
    assert L.savit_colsum_f32(ty.data_ptr(), tout.data_ptr(), M, Nc, Nc, st()) == 0
    assert rel(tout.cpu().numpy(), 1.0 + y.astype(np.float64).sum(0)) < 4e-6
    # d(mean label-smoothed CE)/dlogits (train.py:83-90)
    B, C = 7, 1000
    z, lab = f(B, C) * 2, rng.integers(0, C, B)
    tz, tl, tdz = dev(z), torch.as_tensor(lab, dtype=torch.int32).cuda(), torch.empty(B, C, device="cuda")
    assert L.savit_softmax_xent_grad_f32(tz.data_ptr(), tl.data_ptr(), 0.1, 1.0 / B, tdz.data_ptr(), B, C, st()) == 0
    z64 = z.astype(np.float64)
    sm = np.exp(z64 - z64.max(-1, keepdims=True))
    sm /= sm.sum(-1, keepdims=True)
    tgt = np.full((B, C), 0.1 / C)
    tgt[np.arange(B), lab] += 0.9
    assert rel(tdz.cpu().numpy(), (sm - tgt) / B) < 2e-6


def _flat(tree):
    from oracle import torch_ref

    return {k: v.detach().float().cpu().numpy() for k, v in torch_ref.leaves(tree)}


@pytest.mark.parametrize("name,B,img", [("vit_ti_patch16", 4, 224), ("tiny", 3, 32)])
def test_vit_fp32_train_step_vs_autograd(L, name, B, img):
    """BASELINE config 1 as the reference trains it (simple_train.py:72-90 in fp32): loss, EVERY parameter gradient and the AdamW
    update of one step against fp32 autograd of the oracle (<= 1e-5)."""
    from oracle import torch_ref
    from savit_amd.config import ModelConfig, get_config
    from savit_amd.engine_f32 import ViTEngineF32

    if name == "tiny":
        kw = dict(kind="vit", img_size=32, patch=8, embed_dim=64, num_layers=2, num_heads=2, expand_ratio=4, num_classes=10)
        mc, oc = ModelConfig(**kw), vit_ref.Cfg(**kw)
    else:
        mc, oc = get_config(name), vit_ref.get_cfg(name)
    rng = np.random.default_rng(31)
    params = vit_ref.init_params(oc, seed=9, randomize=True)
    images = rng.standard_normal((B, img, img, 3)).astype(np.float32)
    labels = rng.integers(0, oc.num_classes, B)
    eng = ViTEngineF32(mc, B)
    eng.load_params(params)
    logits = eng.forward(torch.as_tensor(images).cuda())
    loss = float(eng.loss_backward(torch.as_tensor(labels).cuda(), 0.1))
    loss_ref, logits_ref, g_ref = torch_ref.loss_and_grads(params, images, labels, oc, 0.1)
    assert rel(logits.cpu().numpy(), logits_ref) < 2e-5
    assert abs(loss - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    got = _flat(eng.grad_tree()["params"])
    assert set(got) == set(g_ref)
    worst = max((rel(got[k], g_ref[k]), k) for k in got)
    print(f"[fp32 train step {name}] loss {loss:.6f} (oracle {loss_ref:.6f}); worst gradient rel-L2 {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < 1e-5, worst
    # a second backward without zeroing accumulates (gradient accumulation), with it reproduces
    g1 = eng.grads.clone()
    eng.loss_backward(torch.as_tensor(labels).cuda(), 0.1)
    assert rel(eng.grads.cpu().numpy(), g1.cpu().numpy()) < 1e-6
    # optax chain of simple_train.py:25-27: clip_by_global_norm(1.0), adam, add_decayed_weights(1e-4), -lr
    p0, g = eng.params.double().clone(), eng.grads.double().clone()
    eng.optimizer_step(lr=1e-3, weight_decay=1e-4, max_norm=1.0)
    gn = float(g.norm())
    g = g * min(1.0, 1.0 / gn)
    m, v = 0.1 * g, 0.001 * g * g
    upd = (m / 0.1) / ((v / 0.001).sqrt() + 1e-8) + 1e-4 * p0
    assert rel(eng.params.cpu().numpy(), (p0 - 1e-3 * upd).cpu().numpy()) < 1e-6


def test_vit_fp32_long_sequence_forward(L):
    """The 256-token limit of the first fp32 attention kernel is gone: 577 tokens (384 x 384, patch 16), head_dim 64."""
    from savit_amd.config import ModelConfig
    from savit_amd.engine_f32 import ViTEngineF32

    kw = dict(kind="vit", img_size=384, patch=16, embed_dim=128, num_layers=2, num_heads=2, expand_ratio=4, num_classes=10)
    mc, oc = ModelConfig(**kw), vit_ref.Cfg(**kw)
    rng = np.random.default_rng(33)
    params = vit_ref.init_params(oc, seed=2, randomize=True)
    images = rng.standard_normal((2, 384, 384, 3)).astype(np.float32)
    eng = ViTEngineF32(mc, 2)
    eng.load_params(params)
    assert rel(eng.forward(torch.as_tensor(images).cuda()).cpu().numpy(), vit_ref.forward(params, images, oc, mode="f32")) < 2e-5


def test_cait_fp32_default_dtype_vs_oracle(L):
    """create_model('cait_xxs_24') - the reference's default dtype, and the arithmetic its CaiT branch always uses
    (create_model.py:50-213 drop dtype) - within 2e-5 of the fp32 oracle; models/cait_test.py:13-40 shapes."""
    from savit_amd.model import create_model

    model = create_model("cait_xxs_24")
    assert model.dtype == torch.float32
    oc = vit_ref.get_cfg("cait_xxs_24")
    rng = np.random.default_rng(41)
    params = vit_ref.init_params(oc, seed=7, randomize=True)
    images = rng.standard_normal((2, 224, 224, 3)).astype(np.float32)
    logits = model.apply(params, torch.as_tensor(images).cuda(), is_training=False)
    assert logits.dtype == torch.float32 and tuple(logits.shape) == (2, 1000)
    ref32 = vit_ref.forward(params, images, oc, mode="f32")
    ref64 = vit_ref.forward(params, images, oc, mode="f64")
    r32, r64 = rel(logits.cpu().numpy(), ref32), rel(logits.cpu().numpy(), ref64)
    print(f"[fp32 cait_xxs_24] logits rel-L2: engine vs fp32 oracle {r32:.2e}, vs fp64 oracle {r64:.2e} (fp32 oracle vs fp64 {rel(ref32, ref64):.2e})")
    assert r32 < 2e-5 and r64 < 2e-5
    labels = rng.integers(0, 1000, 2)
    loss = float(model.engine(2).loss_fn(torch.as_tensor(labels).cuda(), 0.1))
    assert abs(loss - vit_ref.loss_fn(ref64, labels, 0.1)) < 2e-5 * max(1.0, abs(loss))
    out, _ = create_model("cait_xxs_24").init_with_output(0, torch.ones(2, 224, 224, 3, device="cuda"), is_training=False)
    assert tuple(out.shape) == (2, 1000) and float(out.abs().max()) == 0.0  # zero-init head (cait.py:179-182)


def test_cait_fp32_stochastic_depth_masks(L):
    """Training mode with explicit per-sample keep masks (stochastic_depth.py:16-27) on a small CaiT with 2 class-attention layers."""
    from savit_amd.config import ModelConfig
    from savit_amd.engine_f32 import CaiTEngineF32

    kw = dict(kind="cait", img_size=32, patch=8, embed_dim=64, num_layers=3, num_layers_token_only=2, num_heads=4, expand_ratio=4, num_classes=10,
              stoch_depth_rate=0.3, layerscale_eps=0.1)
    mc, oc = ModelConfig(**kw), vit_ref.Cfg(**kw)
    rng = np.random.default_rng(43)
    params = vit_ref.init_params(oc, seed=6, randomize=True)
    B = 5
    images = rng.standard_normal((B, 32, 32, 3)).astype(np.float32)
    masks = (rng.random((5, 2, B)) < 0.7).astype(np.float32)
    eng = CaiTEngineF32(mc, B)
    eng.load_params(params)
    for training, km in ((False, None), (True, masks)):
        got = eng.forward(torch.as_tensor(images).cuda(), is_training=training, keep_masks=None if km is None else torch.as_tensor(km)).cpu().numpy()
        assert rel(got, vit_ref.forward(params, images, oc, mode="f32", is_training=training, keep_masks=km)) < 2e-5, training
    eng.gen.manual_seed(5)
    a = eng.forward(torch.as_tensor(images).cuda(), is_training=True).clone()
    eng.gen.manual_seed(5)
    assert torch.equal(a, eng.forward(torch.as_tensor(images).cuda(), is_training=True))  # the masks come from the engine's seeded generator


def test_layerscale_bwd_f32(L):
    """savit_layerscale_bwd_f32 (round 6): the VJP of LayerScale x stochastic depth as cait.py:36-52 composes them, against fp64."""
    rng = np.random.default_rng(47)
    for M, d, rps in ((197 * 3, 192, 197), (5, 64, 1), (100, 384, 0)):
        dres, br = rng.standard_normal((M, d)).astype(np.float32), rng.standard_normal((M, d)).astype(np.float32)
        ls = rng.standard_normal(d).astype(np.float32)
        nb = (M + rps - 1) // rps if rps else 0
        rs = (rng.random(nb) < 0.7).astype(np.float32) / 0.7 if rps else None
        rows = np.repeat(rs, rps)[:M].astype(np.float64) if rps else np.ones(M)
        dls0 = rng.standard_normal(d).astype(np.float32)
        t = [dev(x) for x in (dres, br, ls)]
        trs = dev(rs) if rps else None
        out, dls = torch.empty(M, d, device="cuda"), dev(dls0)
        rc = L.savit_layerscale_bwd_f32(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), trs.data_ptr() if rps else None, max(rps, 1), out.data_ptr(),
                                        dls.data_ptr(), M, d, st())
        assert rc == 0
        g = dres.astype(np.float64) * rows[:, None]
        assert rel(out.cpu().numpy(), g * ls[None, :]) < 1e-6
        assert rel(dls.cpu().numpy(), dls0 + (g * br).sum(0)) < 2e-6  # accumulates into dls


@pytest.mark.parametrize("case,B,training", [("small", 5, True), ("small", 5, False), ("cait_xxs_24", 2, False)])
def test_cait_fp32_train_step_vs_autograd(L, case, B, training):
    """Round 6: the fp32 TRAIN step of a CaiT - the arithmetic the reference always uses for this family
    (cait.py:147-154 / create_model.py:115-123 drop dtype) and what train.py:77-100 would differentiate: loss, EVERY parameter gradient
    (talking-heads matrices, LayerScale, class attention included) against fp32 autograd of the oracle, with explicit stochastic-depth
    keep masks in training mode, and the AdamW update."""
    from oracle import torch_ref
    from savit_amd.config import ModelConfig, get_config
    from savit_amd.engine_f32 import CaiTEngineF32

    if case == "small":
        kw = dict(kind="cait", img_size=32, patch=8, embed_dim=64, num_layers=3, num_layers_token_only=2, num_heads=4, expand_ratio=4, num_classes=10,
                  stoch_depth_rate=0.3, layerscale_eps=0.1)
        mc, oc, img = ModelConfig(**kw), vit_ref.Cfg(**kw), 32
    else:
        mc, oc, img = get_config(case), vit_ref.get_cfg(case), 224
    rng = np.random.default_rng(53)
    params = vit_ref.init_params(oc, seed=12, randomize=True)
    images = rng.standard_normal((B, img, img, 3)).astype(np.float32)
    labels = rng.integers(0, oc.num_classes, B)
    masks = (rng.random((oc.num_layers + oc.num_layers_token_only, 2, B)) < 0.7).astype(np.float32) if training else None
    eng = CaiTEngineF32(mc, B)
    eng.load_params(params)
    logits = eng.forward(torch.as_tensor(images).cuda(), is_training=training, keep_masks=None if masks is None else torch.as_tensor(masks)).clone()
    loss = float(eng.loss_backward(torch.as_tensor(labels).cuda(), 0.1))
    assert torch.equal(logits, eng.logits)  # the re-run of the forward that saves activations reproduces it bit for bit
    loss_ref, logits_ref, g_ref = torch_ref.loss_and_grads(params, images, labels, oc, 0.1, is_training=training, keep_masks=masks)
    assert rel(logits.cpu().numpy(), logits_ref) < 2e-5
    assert abs(loss - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    got = _flat(eng.grad_tree()["params"])
    assert set(got) == set(g_ref)
    worst = max((rel(got[k], g_ref[k]), k) for k in got)
    print(f"[fp32 CaiT train step {case} training={training}] loss {loss:.6f} (oracle {loss_ref:.6f}); worst gradient rel-L2 {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < 2e-5, worst
    # a second forward + backward (now saving activations in the forward itself) reproduces the gradients
    g1 = eng.grads.clone()
    eng.forward(torch.as_tensor(images).cuda(), is_training=training, keep_masks=None if masks is None else torch.as_tensor(masks))
    eng.loss_backward(torch.as_tensor(labels).cuda(), 0.1)
    assert rel(eng.grads.cpu().numpy(), g1.cpu().numpy()) < 1e-6
    # optax chain of train.py:25-27 (descent sign of simple_train.py:27): clip_by_global_norm(1.0), adam, add_decayed_weights(1e-4), -lr
    p0, g = eng.params.double().clone(), eng.grads.double().clone()
    eng.optimizer_step(lr=1e-3, weight_decay=1e-4, max_norm=1.0)
    gn = float(g.norm())
    g = g * min(1.0, 1.0 / gn)
    m, v = 0.1 * g, 0.001 * g * g
    upd = (m / 0.1) / ((v / 0.001).sqrt() + 1e-8) + 1e-4 * p0
    assert rel(eng.params.cpu().numpy(), (p0 - 1e-3 * upd).cpu().numpy()) < 1e-6


@pytest.mark.parametrize("case,B", [("tiny", 5), ("p16", 3), ("mixer_s_patch32", 2)])
def test_mixer_fp32_train_step_vs_autograd(L, case, B):
    """The fp32 TRAIN step of an MLP-Mixer - the reference's default arithmetic for this family (mlp_mixer.py:44-64 with dtype=float32,
    differentiated by train.py:77-100): loss and EVERY parameter gradient (token-mixing kernels and their biases included: padded
    storage, pads stay zero) against fp32 autograd of the oracle, then the AdamW update."""
    from oracle import torch_ref
    from savit_amd.config import ModelConfig, get_config
    from savit_amd.engine_f32 import MixerEngineF32

    small = {"tiny": dict(kind="mixer", num_layers=2, num_heads=1, embed_dim=128, patch=8, num_classes=16, img_size=32),
             "p16": dict(kind="mixer", num_layers=2, num_heads=1, embed_dim=128, patch=16, num_classes=104, img_size=224)}
    if case in small:
        mc, oc, img = ModelConfig(**small[case]), vit_ref.Cfg(**small[case]), small[case]["img_size"]
    else:
        mc, oc, img = get_config(case), vit_ref.get_cfg(case), 224
    rng = np.random.default_rng(57)
    params = vit_ref.init_params(oc, seed=13, randomize=True)
    images = rng.standard_normal((B, img, img, 3)).astype(np.float32)
    labels = rng.integers(0, oc.num_classes, B)
    eng = MixerEngineF32(mc, B)
    eng.load_params(params)
    logits = eng.forward(torch.as_tensor(images).cuda(), is_training=True).clone()
    loss = float(eng.loss_backward(torch.as_tensor(labels).cuda(), 0.1))
    assert torch.equal(logits, eng.logits)  # the re-run of the forward that saves activations reproduces it bit for bit
    loss_ref, logits_ref, g_ref = torch_ref.loss_and_grads(params, images, labels, oc, 0.1)
    assert rel(logits.cpu().numpy(), logits_ref) < 2e-5
    assert abs(loss - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    got = _flat(eng.grad_tree()["params"])
    assert set(got) == set(g_ref)
    # the bias of the second token-mixing Dense shifts every channel of a token alike and only LayerNorms over the channels read the
    # stream after it: its gradient is zero up to rounding in the oracle and here - held to an absolute bar, not a relative one
    floor = 1e-6 * max(float(np.linalg.norm(v)) for v in g_ref.values())
    null = [k for k in got if float(np.linalg.norm(g_ref[k])) < floor]
    assert all(k.endswith("FFBlock_0/Dense_1/bias") for k in null) and all(float(np.linalg.norm(got[k])) < floor for k in null), null
    worst = max((rel(got[k], g_ref[k]), k) for k in got if k not in null)
    print(f"[fp32 Mixer train step {case}] loss {loss:.6f} (oracle {loss_ref:.6f}); worst gradient rel-L2 {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < 2e-5, worst
    # the padding of the token-mixing kernels received no gradient
    lay = eng.layout
    for l in range(mc.num_layers):
        for nm, logical in ((f"l{l}.tW1", (mc.n_patches, mc.tokens_hidden)), (f"l{l}.tW2", (mc.tokens_hidden, mc.n_patches))):
            off, shape = lay.off[nm]
            g = eng.grads[off:off + int(np.prod(shape))].view(*shape).clone()
            g[:logical[0], :logical[1]] = 0
            assert float(g.abs().max()) == 0.0, nm
    # a second forward + backward (now saving activations in the forward itself) reproduces the gradients
    g1 = eng.grads.clone()
    eng.forward(torch.as_tensor(images).cuda(), is_training=True)
    eng.loss_backward(torch.as_tensor(labels).cuda(), 0.1)
    assert rel(eng.grads.cpu().numpy(), g1.cpu().numpy()) < 1e-6
    p0, g = eng.params.double().clone(), eng.grads.double().clone()
    eng.optimizer_step(lr=1e-3, weight_decay=1e-4, max_norm=1.0)
    gn = float(g.norm())
    g = g * min(1.0, 1.0 / gn)
    m, v = 0.1 * g, 0.001 * g * g
    upd = (m / 0.1) / ((v / 0.001).sqrt() + 1e-8) + 1e-4 * p0
    assert rel(eng.params.cpu().numpy(), (p0 - 1e-3 * upd).cpu().numpy()) < 1e-6


@pytest.mark.parametrize("case,B", [("tiny", 3), ("two_inner_heads", 2), ("tnt_s_patch16", 2)])
def test_tnt_fp32_train_step_vs_autograd(L, case, B):
    """The fp32 TRAIN step of a TNT (tnt.py:150-193 at create_model's default dtype, differentiated by train.py:77-100): loss and EVERY
    parameter gradient - pixel stream, patch stream and the Inner2Outer that joins them in every layer, both position tables, the
    head-padded inner attention kernels (their pad columns receive exactly zero) - against fp32 autograd of the oracle; AdamW."""
    from oracle import torch_ref
    from savit_amd.config import ModelConfig, get_config
    from savit_amd.engine_f32 import TNTEngineF32

    small = {"tiny": dict(kind="tnt", num_layers=2, num_heads=2, embed_dim=32, patch=16, num_classes=10, img_size=32, inner_num_heads=2,
                          inner_embed_dim=8),
             "two_inner_heads": dict(kind="tnt", num_layers=3, num_heads=4, embed_dim=64, patch=16, num_classes=24, img_size=64, inner_num_heads=2,
                                     inner_embed_dim=24)}
    if case in small:
        mc, oc, img = ModelConfig(**small[case]), vit_ref.Cfg(**small[case]), small[case]["img_size"]
    else:
        mc, oc, img = get_config(case), vit_ref.get_cfg(case), 224
    rng = np.random.default_rng(59)
    params = vit_ref.init_params(oc, seed=14, randomize=True)
    images = rng.standard_normal((B, img, img, 3)).astype(np.float32)
    labels = rng.integers(0, oc.num_classes, B)
    eng = TNTEngineF32(mc, B)
    eng.load_params(params)
    logits = eng.forward(torch.as_tensor(images).cuda(), is_training=True).clone()
    loss = float(eng.loss_backward(torch.as_tensor(labels).cuda(), 0.1))
    assert torch.equal(logits, eng.logits)  # the re-run of the forward that saves activations reproduces it bit for bit
    loss_ref, logits_ref, g_ref = torch_ref.loss_and_grads(params, images, labels, oc, 0.1)
    assert rel(logits.cpu().numpy(), logits_ref) < 2e-5
    assert abs(loss - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    got = _flat(eng.grad_tree()["params"])
    assert set(got) == set(g_ref)
    worst = max((rel(got[k], g_ref[k]), k) for k in got)
    print(f"[fp32 TNT train step {case}] loss {loss:.6f} (oracle {loss_ref:.6f}); worst gradient rel-L2 {worst[0]:.2e} ({worst[1]})")
    assert worst[0] < 2e-5, worst
    # the head padding of the inner attention kernels received no gradient: everything outside the logical corner is exactly zero
    lay, di, Hi = eng.layout, mc.inner_embed_dim, mc.inner_num_heads
    hdi = di // Hi
    for l in range(mc.num_layers):
        gq = lay.view(eng.grads, f"l{l}.iWqkv").view(di, 3, Hi, eng.HDP)[..., hdi:]
        go = lay.view(eng.grads, f"l{l}.iWo").view(Hi, eng.HDP, di)[:, hdi:, :]
        assert float(gq.abs().max()) == 0.0 and float(go.abs().max()) == 0.0, l
    g1 = eng.grads.clone()
    eng.forward(torch.as_tensor(images).cuda(), is_training=True)
    eng.loss_backward(torch.as_tensor(labels).cuda(), 0.1)
    assert rel(eng.grads.cpu().numpy(), g1.cpu().numpy()) < 1e-6
    p0, g = eng.params.double().clone(), eng.grads.double().clone()
    eng.optimizer_step(lr=1e-3, weight_decay=1e-4, max_norm=1.0)
    gn = float(g.norm())
    g = g * min(1.0, 1.0 / gn)
    m, v = 0.1 * g, 0.001 * g * g
    upd = (m / 0.1) / ((v / 0.001).sqrt() + 1e-8) + 1e-4 * p0
    assert rel(eng.params.cpu().numpy(), (p0 - 1e-3 * upd).cpu().numpy()) < 1e-6


def _leaves(t):
    for v in t.values():
        if isinstance(v, dict):
            yield from _leaves(v)
        else:
            yield v
