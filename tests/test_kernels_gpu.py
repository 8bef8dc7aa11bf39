"""Per-kernel parity tests (GPU).  Every kernel is called through the C-ABI (ctypes -> libsavit.so) and
checked against fp32/fp64 CPU math from the oracle on the same seeded, bf16-rounded inputs.

Tolerances (SURVEY A.5 ii): a bf16-output kernel must equal bf16(oracle_fp32) up to rounding-boundary
flips: relative L2 <= 1e-3 (north_star) - in practice ~1e-4 - and max |err| <= 1 bf16 ulp of the value
scale; fp32 outputs: relative L2 <= 2e-5 plus the bf16 rounding the reference applies at that point."""
import numpy as np
import pytest
import torch

from oracle import vit_ref

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import savit_amd  # noqa: F401
    from savit_amd import ops as _ops

    return _ops


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def dev(x, dtype=None):
    t = torch.as_tensor(np.asarray(x)).cuda()
    return t.to(dtype) if dtype is not None else t


def host(t):
    return t.detach().float().cpu().numpy()


def rb(x):  # bf16-rounded fp32 numpy
    return vit_ref.bf16_round(np.asarray(x, np.float32))


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("rows,d", [(1, 192), (5, 192), (197 * 3, 384), (394, 768), (131, 1024), (64, 288), (7, 32),
                                    (16 * 196 * 3 + 5, 24), (1000, 40), (33, 64), (3, 24)])  # d <= 64: the narrow-row kernels (TNT pixels)
def test_layernorm_fwd(ops, rows, d):
    rng = np.random.default_rng(rows * 1000 + d)
    x = (rng.standard_normal((rows, d)) * 2 + 0.3).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    b = (0.1 * rng.standard_normal(d)).astype(np.float32)
    y, mean, rstd = ops.layernorm_fwd(dev(x), dev(g), dev(b))
    ref = vit_ref.layer_norm(vit_ref.Policy("bf16"), x, g, b)  # fp32 stats, bf16 params/out (A.2)
    out = host(y)
    assert rel(out, ref) < 1e-3
    assert np.abs(out - ref).max() <= 2 ** -7 * max(1.0, np.abs(ref).max())
    assert rel(host(mean), x.mean(-1)) < 1e-5
    assert rel(host(rstd), 1 / np.sqrt(x.astype(np.float64).var(-1) + 1e-6)) < 1e-4


def test_layernorm_fwd_strided_rows(ops):
    """cls-row view: x[:, 0, :] of a [B, N, d] residual stream (final LN feeds only row 0: vit.py:95)."""
    rng = np.random.default_rng(0)
    B, N, d = 6, 197, 384
    x = rng.standard_normal((B, N, d)).astype(np.float32)
    g = np.ones(d, np.float32)
    b = np.zeros(d, np.float32)
    xt = dev(x)
    y, _, _ = ops.layernorm_fwd(xt[:, 0, :], dev(g), dev(b))
    ref = vit_ref.layer_norm(vit_ref.Policy("bf16"), x[:, 0, :], g, b)
    assert rel(host(y), ref) < 1e-3


@pytest.mark.parametrize("rows,d,with_res", [(5, 192, False), (197 * 4, 384, True), (394, 768, True), (33, 1024, True),
                                             (16 * 196 * 3 + 5, 24, True), (1000, 40, True), (37, 64, False), (3, 24, True)])
def test_layernorm_bwd(ops, rows, d, with_res):
    rng = np.random.default_rng(rows + d)
    x = (rng.standard_normal((rows, d)) * 1.5 + 0.2).astype(np.float32)
    g = rb(1 + 0.1 * rng.standard_normal(d))
    b = rb(0.1 * rng.standard_normal(d))
    dy = rb(rng.standard_normal((rows, d)))
    dres = rng.standard_normal((rows, d)).astype(np.float32) if with_res else None
    # fp64 autograd reference
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    gt = torch.tensor(g, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    yt = torch.nn.functional.layer_norm(xt, (d,), gt, bt, eps=1e-6)
    yt.backward(torch.tensor(dy, dtype=torch.float64))
    dx_ref = xt.grad.numpy() + (dres if with_res else 0)
    y, mean, rstd = ops.layernorm_fwd(dev(x), dev(g), dev(b))
    dgamma = torch.zeros(d, device="cuda")
    dbeta = torch.zeros(d, device="cuda")
    dcol = torch.zeros(d, device="cuda")
    dxb = torch.empty((rows, d), dtype=bf16, device="cuda")
    dx = ops.layernorm_bwd(dev(dy, bf16), dev(x), dev(g), mean, rstd, dgamma, dbeta,
                           dres_in=None if dres is None else dev(dres), dx_bf16=dxb, dcolsum=dcol)
    assert rel(host(dx), dx_ref) < 2e-5
    assert rel(host(dxb), rb(dx_ref)) < 1e-3
    assert rel(host(dgamma), gt.grad.numpy()) < 1e-4
    assert rel(host(dbeta), bt.grad.numpy()) < 1e-4
    assert rel(host(dcol), dx_ref.sum(0)) < 1e-4 or np.abs(host(dcol) - dx_ref.sum(0)).max() < 1e-3


# ------------------------------------------------------------------------------------------ GEMM TN
def _mk(rng, M, K, scale=1.0):
    return rb(rng.standard_normal((M, K)) * scale)


# every tile the product library holds (= every tile the auto heuristic can pick: tests/test_abi.py); the earlier rounds' other forms
# exist in SAVIT_EXPERIMENTS builds only
PRODUCT_TILES = [6, 12, 13, 17, 18, 20, 21, 22, 24]  # (24: the few-rows kernel - no LDS; any M, sized for M <= 256)  # (22 on these small shapes = 21: the persistent grid needs more tiles than CUs - see test_gemm_persistent_320 below)


@pytest.mark.parametrize("tile", PRODUCT_TILES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 256, 128), (197 * 2, 192 * 3, 192), (591, 768, 768), (1000, 384, 1536),
                                   (37, 1000, 192), (300, 64, 128)])
def test_gemm_bf16_plain(ops, tile, M, N, K):
    rng = np.random.default_rng(M + N + K)
    A, Bt = _mk(rng, M, K), _mk(rng, N, K, 1 / np.sqrt(K))
    C = torch.full((M, N), float("nan"), dtype=bf16, device="cuda")
    ops.gemm_tn(dev(A, bf16), dev(Bt, bf16), C, 0, tile=tile)
    ref = A.astype(np.float64) @ Bt.astype(np.float64).T
    out = host(C)
    assert np.isfinite(out).all()
    assert rel(out, rb(ref)) < 1e-3, rel(out, rb(ref))


def test_gemm_identity_asymmetric(ops):
    """A = I with an asymmetric B catches row/col swaps in the C-write (guide section 3)."""
    K = 128
    A = np.eye(K, dtype=np.float32)
    Bt = rb(np.arange(K * K, dtype=np.float32).reshape(K, K) % 251 - 100)
    C = torch.zeros((K, K), dtype=bf16, device="cuda")
    ops.gemm_tn(dev(A, bf16), dev(Bt, bf16), C, 0)
    assert np.array_equal(host(C), rb(Bt.T))


def test_gemm_qkv_alpha_and_strided_views(ops):
    rng = np.random.default_rng(5)
    M, d, H = 197 * 2, 384, 6
    x, W = _mk(rng, M, d), _mk(rng, 3 * d, d, 1 / np.sqrt(d))
    C = torch.zeros((M, 3 * d), dtype=bf16, device="cuda")
    ops.gemm_tn(dev(x, bf16), dev(W, bf16), C, 0, alpha=1 / 8.0, alpha_cols=d)
    ref = x.astype(np.float64) @ W.astype(np.float64).T
    ref[:, :d] /= 8.0
    assert rel(host(C), rb(ref)) < 1e-3
    # A as a column slice of a wider buffer (lda > K), C as a column slice too
    wide = torch.zeros((M, 3 * d), dtype=bf16, device="cuda")
    wide[:, d:2 * d] = dev(x, bf16)
    Cw = torch.zeros((M, 4 * d), dtype=bf16, device="cuda")
    ops.gemm_tn(wide[:, d:2 * d], dev(W[:d], bf16), Cw[:, d:2 * d], 0)
    assert rel(host(Cw[:, d:2 * d]), rb(x.astype(np.float64) @ W[:d].astype(np.float64).T)) < 1e-3
    assert float(Cw[:, :d].abs().max()) == 0 and float(Cw[:, 2 * d:].abs().max()) == 0


@pytest.mark.parametrize("tile", PRODUCT_TILES)
def test_gemm_bias_gelu(ops, tile):
    rng = np.random.default_rng(6)
    M, d, F = 197 * 3, 192, 768
    x, W = _mk(rng, M, d), _mk(rng, F, d, 1 / np.sqrt(d))
    b = (0.1 * rng.standard_normal(F)).astype(np.float32)
    U = torch.zeros((M, F), dtype=bf16, device="cuda")
    Aact = torch.zeros((M, F), dtype=bf16, device="cuda")
    ops.gemm_tn(dev(x, bf16), dev(W, bf16), U, 1, C2=Aact, bias=dev(b), tile=tile)
    u_ref = rb(x.astype(np.float64) @ W.astype(np.float64).T + rb(b))
    assert rel(host(U), u_ref) < 1e-3
    pol = vit_ref.Policy("f32")
    a_ref = rb(vit_ref.gelu_tanh(pol, host(U)))  # gelu of the kernel's own rounded u
    assert rel(host(Aact), a_ref) < 1e-3
    assert np.abs(host(Aact) - a_ref).max() <= 2 ** -7 * max(1.0, np.abs(a_ref).max())


@pytest.mark.parametrize("tile", PRODUCT_TILES)
def test_gemm_residual_layerscale_stochdepth(ops, tile):
    rng = np.random.default_rng(7)
    B, N, d, F = 3, 197, 192, 768
    M = B * N
    a, W = _mk(rng, M, F), _mk(rng, d, F, 1 / np.sqrt(F))
    bias = (0.1 * rng.standard_normal(d)).astype(np.float32)
    res = rng.standard_normal((M, d)).astype(np.float32)
    ls = (0.5 + rng.random(d)).astype(np.float32)
    keep = np.array([1 / 0.9, 0.0, 1 / 0.9], np.float32)
    out = torch.zeros((M, d), dtype=torch.float32, device="cuda")
    ops.gemm_tn(dev(a, bf16), dev(W, bf16), out, 2, bias=dev(bias), aux=dev(res), colscale=dev(ls), rowscale=dev(keep),
                rows_per_sample=N, tile=tile)
    branch = rb(a.astype(np.float64) @ W.astype(np.float64).T + rb(bias))
    ref = res + np.repeat(keep, N)[:, None] * ls[None, :] * branch
    assert rel(host(out), ref) < 1e-3
    assert np.array_equal(host(out)[N:2 * N], res[N:2 * N])  # dropped sample: residual only
    out2 = torch.zeros((M, d), dtype=torch.float32, device="cuda")
    ops.gemm_tn(dev(a, bf16), dev(W, bf16), out2, 2, aux=dev(res), tile=tile)  # plain ViT form, no bias
    assert rel(host(out2), res + rb(a.astype(np.float64) @ W.astype(np.float64).T)) < 2e-4


def test_gemm_dgelu_and_colsum(ops):
    rng = np.random.default_rng(8)
    M, d, F = 197 * 2 + 5, 192, 768
    dy, W2 = _mk(rng, M, d), _mk(rng, F, d, 1 / np.sqrt(d))  # d_a = dy @ W2^T as TN: Bt = W2 [F, d]
    u = _mk(rng, M, F)
    dU = torch.zeros((M, F), dtype=bf16, device="cuda")
    cs = torch.zeros(F, device="cuda")
    ops.gemm_tn(dev(dy, bf16), dev(W2, bf16), dU, 3, aux=dev(u, bf16), colsum=cs)
    # the reference's bf16 graph materialises d_a = dy @ W2^T in bf16 (cotangent of the bf16 gelu output) and only then
    # multiplies by gelu'(u) and rounds again; the kernel rounds at the same two points.
    ut = torch.tensor(u, dtype=torch.float64, requires_grad=True)
    d_a = rb(dy.astype(np.float64) @ W2.astype(np.float64).T)
    torch.nn.functional.gelu(ut, approximate="tanh").backward(torch.tensor(d_a.astype(np.float64)))
    ref = ut.grad.numpy()
    assert rel(host(dU), rb(ref)) < 1e-3
    assert rel(host(cs), host(dU).astype(np.float64).sum(0)) < 1e-4


@pytest.mark.parametrize("tile", [0, 6, 12, 13, 17, 20, 21])
def test_gemm_dgelu_colsum_slab_is_deterministic(ops, tile):
    """colsum as a [rows, N] slab of per-row-tile partials + savit_colsum_finalize: same sums as the atomic form, and bitwise
    reproducible (no atomics)."""
    rng = np.random.default_rng(18)
    M, d, F = 197 * 3 + 11, 192, 768
    dy, W2, u = _mk(rng, M, d), _mk(rng, F, d, 1 / np.sqrt(d)), _mk(rng, M, F)
    rows = ops.gemm_colsum_rows(M, F, d, tile)
    assert rows >= 2 * ((M + 255) // 256)
    outs = []
    for _ in range(2):
        dU = torch.zeros((M, F), dtype=bf16, device="cuda")
        slab = torch.full((rows, F), float("nan"), device="cuda")
        ops.gemm_tn(dev(dy, bf16), dev(W2, bf16), dU, 3, aux=dev(u, bf16), colsum=slab, tile=tile)
        assert torch.isfinite(slab).all()  # every slab entry is written
        cs = torch.full((F,), 2.0, device="cuda")
        ops.colsum_finalize(slab, cs, accumulate=True)
        outs.append(cs.clone())
        assert rel(host(cs) - 2.0, host(dU).astype(np.float64).sum(0)) < 1e-4
    assert torch.equal(outs[0], outs[1])
    cs0 = torch.full((F,), 7.0, device="cuda")
    ops.colsum_finalize(slab, cs0, accumulate=False)
    assert torch.equal(cs0, outs[0] - 2.0) or rel(host(cs0), host(outs[0]) - 2.0) < 1e-6
    with pytest.raises(ValueError):
        ops.gemm_tn(dev(dy, bf16), dev(W2, bf16), dU, 3, aux=dev(u, bf16), colsum=slab[:-1], tile=tile)


def _epi_case(ops, rng_seed, M, N, K, epi, tile, plain_bias=True):
    """One launch of every fused epilogue on seeded operands; returns (outputs, fp64 reference of the main output)."""
    g = torch.Generator(device="cuda").manual_seed(rng_seed)
    A = torch.randn(M, K, device="cuda", generator=g).to(bf16)
    Bt = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(bf16)
    bias = 0.1 * torch.randn(N, device="cuda", generator=g)
    ref = (A.double() @ Bt.double().T).cpu().numpy()
    bias_r = rb(bias.cpu().numpy())
    kw, outs = {}, []
    if epi == 0 and not plain_bias:  # the qkv form: no bias, the first third of the columns scaled (attention.py:39)
        C = torch.full((M, N), float("nan"), device="cuda", dtype=bf16)
        kw = dict(alpha=0.125, alpha_cols=N // 3 // 128 * 128)
        want = ref.copy()
        want[:, :kw["alpha_cols"]] *= 0.125
        want = rb(want)
    elif epi == 0:
        C = torch.full((M, N), float("nan"), device="cuda", dtype=bf16)
        kw = dict(bias=bias)
        want = rb(ref + bias_r)
    elif epi == 1:
        C = torch.full((M, N), float("nan"), device="cuda", dtype=bf16)
        kw = dict(bias=bias, C2=torch.full((M, N), float("nan"), device="cuda", dtype=bf16))
        want = rb(ref + bias_r)
    elif epi == 2:
        C = torch.full((M, N), float("nan"), device="cuda")
        aux = torch.randn(M, N, device="cuda", generator=g)
        kw = dict(bias=bias, aux=aux)
        want = aux.cpu().numpy().astype(np.float64) + rb(ref + bias_r)
    elif epi == 3:
        C = torch.full((M, N), float("nan"), device="cuda", dtype=bf16)
        u = torch.randn(M, N, device="cuda", generator=g).to(bf16)
        rows = ops.gemm_colsum_rows(M, N, K, tile)
        kw = dict(aux=u, colsum=torch.full((rows, N), float("nan"), device="cuda"))
        ut = u.double().cpu().requires_grad_(True)
        torch.nn.functional.gelu(ut, approximate="tanh").backward(torch.tensor(rb(ref).astype(np.float64)))
        want = rb(ut.grad.numpy())
    else:
        C = torch.full((M, N), float("nan"), device="cuda")
        kw = dict(bias=bias)
        want = ref + bias_r
    ops.gemm_tn(A, Bt, C, epi, tile=tile, **kw)
    outs = [C] + [kw[k] for k in ("C2", "colsum") if k in kw]
    return outs, want


@pytest.mark.parametrize("epi", [0, 1, 2, 4])
@pytest.mark.parametrize("M,N,K", [(128, 768, 3072), (128, 3072, 768), (256, 384, 1536), (37, 1000, 768), (4, 768, 768), (300, 64, 96)])
def test_gemm_few_rows_kernel_is_bitwise_the_lds_tiles(ops, epi, M, N, K):
    """Tile 24 (round 5: one wave per 16 x 16 output tile over all of K, fragments straight from global memory - the B cls rows of a
    ViT's last layer and head, CaiT's class-attention layers) against exact fp64 math and BITWISE against the LDS tile that served
    those shapes before (12; K % 64 != 0: the ring, 6): same K order per output element, so a half batch on this kernel and a whole
    batch on the LDS tiles still agree to fp32 summation order of the REDUCTIONS only (tests/test_ddp_gpu.py).  The heuristic takes it for
    M <= 256 and every epilogue but GELU' (column-sum slab) and the patch gather."""
    from savit_amd import lib as _lib

    L = _lib.load()
    assert L.savit_gemm_tn_auto_tile_cus(min(M, 256), N, K, epi, 0) == 24 and L.savit_gemm_tn_auto_tile_cus(257, N, K, epi, 0) != 24
    assert L.savit_gemm_tn_auto_tile_cus(128, N, K, 3, 0) != 24
    other = 12 if K % 64 == 0 else 6
    got = {}
    for tile in (24, other):
        outs, want = _epi_case(ops, 77 * epi + K, M, N, K, epi, tile)
        assert all(torch.isfinite(o.float()).all() for o in outs), (tile, "an output element was not written")
        e = rel(host(outs[0]), want)
        assert e < (2e-5 if epi == 4 else 1e-3), (tile, e)
        got[tile] = outs
    for x, y in zip(got[24], got[other]):
        assert torch.equal(x, y)


def test_gemm_few_rows_kernel_takes_any_row_pitch(ops):
    """The cls rows of a dense [B * N, F] activation are B rows N * F elements apart (ViT-L at 1 025 tokens, 256 images: 2.1 GB from the
    first to the last row - more than one 32-bit buffer descriptor spans): the few-rows kernel's descriptors start at each wave's own rows."""
    M, N, K, pitch = 40, 96, 128, 29_000_000  # 39 * 29e6 * 2 B = 2.26 GB between the first and the last row
    big = torch.empty(M * pitch, dtype=bf16, device="cuda")
    A = big.as_strided((M, K), (pitch, 1))
    g = torch.Generator(device="cuda").manual_seed(5)
    A.copy_(torch.randn(M, K, device="cuda", generator=g).to(bf16))
    Bt = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(bf16)
    C = torch.full((M, N), float("nan"), dtype=bf16, device="cuda")
    ops.gemm_tn(A, Bt, C, 0, tile=24)
    ref = (A.double() @ Bt.double().T).cpu().numpy()
    assert torch.isfinite(C.float()).all() and rel(host(C), rb(ref)) < 1e-3
    C0 = torch.full((M, N), float("nan"), dtype=bf16, device="cuda")
    ops.gemm_tn(A, Bt, C0, 0)  # the auto choice for 40 rows is the same kernel
    assert torch.equal(C, C0)
    del big


@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(19700, 768, 128), (19700, 1536, 192), (3001, 1000, 256), (25216, 2304, 64)])
def test_gemm_large_grid_tiles_vs_oracle_and_each_other(ops, epi, M, N, K):
    """The tiles the DeiT-B / ViT-L steps actually run - 18 (192x128 with the last partial round cut into 128-row tiles), 20 (the
    256x256 ping-pong kernel) and 21 (its 320-row form) - on grids of more than one round of workgroups (618 / 1236 tiles of 192x128, 462 / 891 of 256x256:
    tail plans and the grouped tile order are exercised), every fused epilogue: each against exact fp64 math on the same bf16 operands,
    and BITWISE against the plain kernels they replace (17 and 13: same K order per output element)."""
    if epi == 3 and N % 8:
        pytest.skip("column sums need N % 8 == 0")
    got = {}
    for tile in (17, 18, 13, 20, 21):
        if epi == 3 and tile == 18:
            continue  # the tail-split launcher hands column-sum epilogues to the plain kernel (slab rows are per row tile)
        outs, want = _epi_case(ops, 1000 * epi + K, M, N, K, epi, tile)
        assert all(torch.isfinite(o.float()).all() for o in outs), (tile, "an output element was not written")
        e = rel(host(outs[0]), want)
        assert e < (2e-5 if epi == 4 else 1e-3), (tile, e)  # fp32 output: summation order only; bf16-valued ones: rounding flips
        got[tile] = outs
    pairs = [(17, 18), (13, 20), (13, 21)] if epi != 3 else [(13, 20), (13, 21)]
    for a, b in pairs:
        for i, (x, y) in enumerate(zip(got[a], got[b])):
            if epi == 3 and i == 1:
                assert rel(host(x).sum(0), host(y).sum(0)) < 1e-6  # slabs have different row counts per tile: compare the sums
            else:
                assert torch.equal(x, y), (a, b, i)
    if epi != 3:
        assert torch.equal(got[17][0], got[13][0])  # 192x128 and 256x256 tiles also agree bit for bit (same K order)


def test_gemm_f32_head(ops):
    rng = np.random.default_rng(9)
    B, d, C = 37, 192, 1000
    z, Wh = _mk(rng, B, d), _mk(rng, C, d, 1 / np.sqrt(d))
    b = (0.1 * rng.standard_normal(C)).astype(np.float32)
    out = torch.zeros((B, C), dtype=torch.float32, device="cuda")
    ops.gemm_tn(dev(z, bf16), dev(Wh, bf16), out, 4, bias=dev(b), round_out_bf16=True)
    ref = rb(z.astype(np.float64) @ Wh.astype(np.float64).T + rb(b))
    assert rel(host(out), ref) < 1e-3
    ops.gemm_tn(dev(z, bf16), dev(Wh, bf16), out, 4, bias=dev(b), round_out_bf16=False, round_bias_bf16=False)
    assert rel(host(out), z.astype(np.float64) @ Wh.astype(np.float64).T + b) < 2e-5


@pytest.mark.parametrize("img,P,d,tok_off", [(32, 8, 64, 1), (224, 16, 192, 1), (64, 32, 128, 0), (224, 16, 384, 0)])
def test_patch_embed_gemm(ops, img, P, d, tok_off):
    rng = np.random.default_rng(img + P)
    B = 3
    n = (img // P) ** 2
    tokens = n + tok_off
    images = rb(rng.standard_normal((B, img, img, 3)))
    Wpe = _mk(rng, d, P * P * 3, 1 / np.sqrt(P * P * 3))  # [d, K] = kernel^T
    pos = (0.02 * rng.standard_normal((tokens, d))).astype(np.float32)
    x0 = torch.full((B * tokens, d), 7.0, dtype=torch.float32, device="cuda")
    ops.gemm_tn(dev(images, bf16), dev(Wpe, bf16), x0, 5, aux=dev(pos), patch_geom=(img, P, tokens, tok_off))
    tok = vit_ref.patchify(images, P, P).astype(np.float64) @ Wpe.astype(np.float64).T  # [B, n, d]
    ref = rb(tok) + pos[None, tok_off:, :]
    out = host(x0).reshape(B, tokens, d)
    assert rel(out[:, tok_off:], ref) < 1e-3
    if tok_off:
        assert np.all(out[:, 0] == 7.0)  # cls row untouched


# ------------------------------------------------------------------------------------------ wgrad
@pytest.mark.parametrize("M,Kin,Nout,splits", [(64, 128, 128, 1), (197 * 2, 192, 576, 0), (1000, 384, 384, 3), (197 * 3 + 1, 768, 192, 0),
                                               (37, 192, 1000, 0), (2000, 1536, 384, 0)])
def test_wgrad(ops, M, Kin, Nout, splits):
    rng = np.random.default_rng(M + Kin)
    X, dY = _mk(rng, M, Kin), _mk(rng, M, Nout)
    dW = torch.zeros((Kin, Nout), dtype=torch.float32, device="cuda")
    ops.gemm_wgrad(dev(X, bf16), dev(dY, bf16), dW, splits=splits)
    ref = X.astype(np.float64).T @ dY.astype(np.float64)
    assert rel(host(dW), ref) < 2e-5, rel(host(dW), ref)
    ops.gemm_wgrad(dev(X, bf16), dev(dY, bf16), dW, splits=splits)  # accumulates
    assert rel(host(dW), 2 * ref) < 2e-5


@pytest.mark.parametrize("M,Kin,Nout,splits", [(25216, 768, 768, 0), (3000, 768, 3072, 5), (197 * 4, 192, 576, 0), (1000, 256, 1000, 3),
                                               (4096, 64, 192, 9), (70, 768, 768, 0)])
def test_wgrad_slab_reduction(ops, M, Kin, Nout, splits):
    """savit_gemm_bf16_wgrad_ws: partial slabs + ordered sum instead of fp32 atomics - same values, accumulating, and bitwise
    reproducible (the atomic form depends on arrival order)."""
    rng = np.random.default_rng(M + Nout)
    X, dY = dev(_mk(rng, M, Kin), bf16), dev(_mk(rng, M, Nout), bf16)
    ws = ops.wgrad_workspace(M, Kin, Nout, splits)
    base = torch.randn((Kin, Nout), dtype=torch.float32, device="cuda")
    dW = base.clone()
    ops.gemm_wgrad(X, dY, dW, splits=splits, workspace=ws)
    ref = host(X).astype(np.float64).T @ host(dY).astype(np.float64)
    assert rel(host(dW) - host(base), ref) < 2e-5
    again = base.clone()
    ops.gemm_wgrad(X, dY, again, splits=splits, workspace=ws)
    assert torch.equal(dW, again)
    ops.gemm_wgrad(X, dY, dW, splits=splits, workspace=ws)  # accumulates
    assert rel(host(dW) - host(base), 2 * ref) < 2e-5
    # a workspace that is too small falls back to the atomic form (same values)
    small = torch.empty(16, dtype=torch.uint8, device="cuda")
    dW2 = base.clone()
    ops.gemm_wgrad(X, dY, dW2, splits=splits, workspace=small)
    assert rel(host(dW2) - host(base), ref) < 2e-5


@pytest.mark.parametrize("tile", [256, 128, 384, 640])
def test_wgrad_grouped_matches_fp64_and_is_reproducible(ops, tile):
    """savit_gemm_bf16_wgrad_grouped: several weight gradients in one launch, one workgroup per output tile over ALL tokens.  Each
    dW: the fp64 product within fp32 summation error, accumulated onto its old value, bitwise repeatable; ragged Kin / Nout (tiles
    that hang over the matrix edge), a column slice of a wider dY (lddy > Nout), different M per problem."""
    rng = np.random.default_rng(tile)
    shapes = [(1999, 768, 1024), (1999, 256, 256), (3001, 192, 576), (517, 384, 1000), (64, 128, 128), (2048, 1024, 256)]
    probs, refs, bases = [], [], []
    for i, (M, Kin, Nout) in enumerate(shapes):
        X = dev(_mk(rng, M, Kin), bf16)
        wide = dev(_mk(rng, M, Nout + 64), bf16)
        dY = wide[:, 32:32 + Nout] if i % 2 else dev(_mk(rng, M, Nout), bf16)
        base = torch.randn((Kin, Nout), dtype=torch.float32, device="cuda")
        probs.append((X, dY, base.clone()))
        bases.append(base)
        refs.append(host(X).astype(np.float64).T @ host(dY).astype(np.float64))
    ops.gemm_wgrad_grouped(probs, tile=tile)
    for (X, dY, dW), base, ref in zip(probs, bases, refs):
        assert rel(host(dW) - host(base), ref) < 2e-5
    again = [(X, dY, base.clone()) for (X, dY, _), base in zip(probs, bases)]
    ops.gemm_wgrad_grouped(again, tile=tile)
    for (_, _, a), (_, _, b) in zip(probs, again):
        assert torch.equal(a, b)
    ops.gemm_wgrad_grouped(probs, tile=tile)  # accumulates
    for (X, dY, dW), base, ref in zip(probs, bases, refs):
        assert rel(host(dW) - host(base), 2 * ref) < 2e-5
    # a weight's tiles cut over two launches (what the engine does to make every launch an exact multiple of the CU count)
    X, dY, base = probs[0][0], probs[0][1], bases[0]
    nt = ops._lib.load().savit_gemm_wgrad_group_tiles(768, 1024, tile)
    halves = base.clone()
    ops.gemm_wgrad_grouped([(X, dY, halves, 0, 5)], tile=tile)
    ops.gemm_wgrad_grouped([(X, dY, halves, 5, nt - 5), (probs[1][0], probs[1][1], bases[1].clone())], tile=tile)
    whole = base.clone()
    ops.gemm_wgrad_grouped([(X, dY, whole)], tile=tile)
    assert torch.equal(halves, whole)
    with pytest.raises(ValueError):
        ops.gemm_wgrad_grouped([(X, dY, halves, 5, nt)], tile=tile)  # range past the last tile
    many = [(X_, dY_, b_.clone()) for _ in range(8) for (X_, dY_, _), b_ in zip(probs, bases)]  # 48 entries in one launch (narrow models)
    ops.gemm_wgrad_grouped(many, tile=tile)
    for k in range(len(probs)):
        assert torch.equal(many[k][2], again[k][2]) and torch.equal(many[k + 6 * 7][2], again[k][2])
    with pytest.raises(ValueError):
        ops.gemm_wgrad_grouped(probs * 11, tile=tile)  # more than 64 problems
    with pytest.raises(ValueError):
        ops.gemm_wgrad_grouped(probs[:1], tile=192)


def test_wgrad_padded_dy_columns(ops):
    """Head: dlogits lives in a [B, 1024] buffer, dW is [d, 1000]."""
    rng = np.random.default_rng(11)
    B, d, C, Cp = 130, 192, 1000, 1024
    z = _mk(rng, B, d)
    dl = np.zeros((B, Cp), np.float32)
    dl[:, :C] = _mk(rng, B, C)
    dW = torch.zeros((d, C), dtype=torch.float32, device="cuda")
    ops.gemm_wgrad(dev(z, bf16), dev(dl, bf16), dW)
    assert rel(host(dW), z.astype(np.float64).T @ dl[:, :C].astype(np.float64)) < 2e-5


@pytest.mark.parametrize("img,P,d,tok_off", [(32, 8, 64, 1), (224, 16, 192, 1), (64, 32, 128, 0)])
def test_wgrad_patch(ops, img, P, d, tok_off):
    rng = np.random.default_rng(img)
    B = 2
    n = (img // P) ** 2
    tokens = n + tok_off
    images = rb(rng.standard_normal((B, img, img, 3)))
    dx0 = _mk(rng, B * tokens, d)
    dW = torch.zeros((P * P * 3, d), dtype=torch.float32, device="cuda")
    ops.gemm_wgrad(dev(images, bf16), dev(dx0, bf16), dW, patch_geom=(img, P, tokens, tok_off))
    patches = vit_ref.patchify(images, P, P).reshape(B * n, -1).astype(np.float64)
    dy = dx0.reshape(B, tokens, d)[:, tok_off:].reshape(B * n, d).astype(np.float64)
    assert rel(host(dW), patches.T @ dy) < 2e-5


def test_contract_violations_raise(ops):
    A = torch.zeros((64, 80), dtype=bf16, device="cuda")  # K % 32 != 0
    with pytest.raises(ValueError):
        ops.gemm_tn(A, torch.zeros((64, 80), dtype=bf16, device="cuda"), torch.zeros((64, 64), dtype=bf16, device="cuda"), 0)
    with pytest.raises(ValueError):
        ops.layernorm_fwd(torch.zeros((4, 64)), torch.zeros(64), torch.zeros(64))  # CPU tensors: no CPU path


# ------------------------------------------------------------------------------------------ attention
def _attn_ref(qkv, B, N, H, hd=64):
    """fp64 reference on the bf16-rounded inputs; q columns are already scaled."""
    d = H * hd
    x = qkv.astype(np.float64).reshape(B, N, 3, H, hd)
    q, k, v = x[:, :, 0], x[:, :, 1], x[:, :, 2]
    s = np.einsum("bqhd,bkhd->bhqk", q, k)
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    l = e.sum(-1, keepdims=True)
    p = e / l
    o = np.einsum("bhqk,bkhd->bqhd", p, v).reshape(B * N, d)
    return o, (m + np.log(l))[..., 0], p


def _attn_emul(qkv, B, N, H, hd=64, d_o=None, o_saved=None, dq_scale=1.0, chunk=None):
    """Exact (fp64) math WITH the bf16 roundings csrc/attention.hip applies to its MFMA operands - so that what is left between this
    and the kernel is summation order and rounding-boundary flips of the bf16 outputs, not a rounding policy:
      forward : P operand = bf16(exp(S - max)), row sum l from the UNROUNDED exponentials, O = (P V) / l.  chunk = keys per online-
                softmax step of the general kernel (128; None = the resident kernel: one pass, final max): there the exponentials
                of chunk c are rounded relative to the running max after chunk c and rescaled in fp32 afterwards;
      backward: P = exp(S - LSE), delta = rowsum(dO * O_saved) with the bf16 O the forward stored, dP = dO V^T,
                dS = P (dP - delta); operands bf16(P) for dV = P^T dO and bf16(dS) for dQ = dS K * dq_scale, dK = dS^T Q.
    Returns o [B*N, d] (and dqkv [B*N, 3d] when d_o is given), unrounded."""
    d = H * hd
    x = qkv.astype(np.float64).reshape(B, N, 3, H, hd)
    q, k, v = (x[:, :, i].transpose(0, 2, 1, 3) for i in range(3))  # [B, H, N, hd]
    s = q @ k.transpose(0, 1, 3, 2)
    m_fin = s.max(-1, keepdims=True)
    l = np.exp(s - m_fin).sum(-1, keepdims=True)
    if chunk is None:
        p_op = rb(np.exp(s - m_fin)).astype(np.float64)
    else:
        p_op = np.empty_like(s)
        m_run = np.full_like(m_fin, -np.inf)
        for c0 in range(0, N, chunk):
            m_run = np.maximum(m_run, s[..., c0:c0 + chunk].max(-1, keepdims=True))
            p_op[..., c0:c0 + chunk] = rb(np.exp(s[..., c0:c0 + chunk] - m_run)).astype(np.float64) * np.exp(m_run - m_fin)
    o = ((p_op @ v) / l).transpose(0, 2, 1, 3).reshape(B * N, d)
    if d_o is None:
        return o
    do = d_o.astype(np.float64).reshape(B, N, H, hd).transpose(0, 2, 1, 3)
    osv = o_saved.astype(np.float64).reshape(B, N, H, hd).transpose(0, 2, 1, 3)
    p = np.exp(s - m_fin) / l
    delta = (do * osv).sum(-1, keepdims=True)
    ds = p * (do @ v.transpose(0, 1, 3, 2) - delta)
    p_b, ds_b = rb(p).astype(np.float64), rb(ds).astype(np.float64)
    dq = (ds_b @ k) * dq_scale
    dk = ds_b.transpose(0, 1, 3, 2) @ q
    dv = p_b.transpose(0, 1, 3, 2) @ do
    g = np.stack([t.transpose(0, 2, 1, 3) for t in (dq, dk, dv)], axis=2).reshape(B * N, 3 * d)
    return o, g


@pytest.mark.parametrize("B,N,H", [(2, 197, 3), (3, 196, 6), (1, 50, 12), (2, 32, 1), (1, 256, 2), (2, 17, 2), (1, 1, 1),
                                   (37, 197, 12), (90, 100, 3)])  # (the last two: more items than CUs - persistent workgroups with 1 and 2 items)
def test_attention_fwd(ops, B, N, H):
    rng = np.random.default_rng(B * 100 + N)
    d = H * 64
    qkv = rb(rng.standard_normal((B * N, 3 * d)))
    qkv[:, :d] = rb(qkv[:, :d] / 8.0 * 3.0)  # pre-scaled queries, with a spread that makes softmax peaky
    o, lse = ops.attention_fwd(dev(qkv, bf16), B, N, H)
    o_ref, lse_ref, _ = _attn_ref(qkv, B, N, H)
    assert np.isfinite(host(o)).all()
    assert rel(host(o), o_ref) < 3e-3, rel(host(o), o_ref)  # vs exact math: P is rounded to bf16 as an MFMA operand (measured 1.6e-3)
    # vs exact math with that one operand rounding emulated: what is left is fp32 summation order + flips of the bf16 output rounding
    e = rel(host(o), rb(_attn_emul(qkv, B, N, H)))
    assert e < 6e-4, e
    assert np.abs(host(lse) - lse_ref).max() < 1e-4 * max(1.0, np.abs(lse_ref).max())


def test_attention_fwd_spike(ops):
    """One key dominating one query (rule 26: force the data-dependent path; here: large max, masked tail)."""
    rng = np.random.default_rng(3)
    B, N, H = 1, 197, 2
    d = H * 64
    qkv = rb(rng.standard_normal((B * N, 3 * d)) * 0.5)
    qkv[5, :64] = 4.0
    qkv[190, d:d + 64] = 4.0  # key 190 matches query 5 strongly: s = 1024
    o, lse = ops.attention_fwd(dev(qkv, bf16), B, N, H)
    o_ref, lse_ref, _ = _attn_ref(qkv, B, N, H)
    assert np.isfinite(host(o)).all()
    assert rel(host(o), o_ref) < 3e-3
    assert abs(host(lse)[0, 0, 5] - lse_ref[0, 0, 5]) < 1e-2


@pytest.mark.parametrize("B,N,H,hd", [(2, 197, 3, 48), (1, 196, 8, 48), (1, 577, 2, 64), (1, 300, 1, 48), (2, 17, 2, 48), (1, 608, 1, 64),
                                      (392, 16, 4, 16), (5, 16, 4, 32), (2, 50, 2, 16),  # TNT's inner transformer: 16 pixel tokens, padded heads
                                      (1, 256, 2, 48), (2, 129, 2, 48), (1, 384, 1, 64), (1, 545, 1, 32),  # key tiles: 8 = 2 chunks, 5 = 1 + 1, 12, 18 = 4 + 2
                                      # N > 608: the streaming kernels (256-row segments, 8 row blocks per workgroup): ViT-B/16 at 512^2 =
                                      # 1 025 tokens (33 tiles = 4 segments + 1 tile, 5 rounds of query blocks, the last with one wave active),
                                      # 609 (just over the resident limit), 640 / 897 / 1 280 with the other head widths
                                      (1, 1025, 2, 64), (1, 609, 1, 64), (2, 640, 3, 48), (1, 897, 1, 32), (1, 1280, 2, 16)])
def test_attention_general_fwd_bwd(ops, B, N, H, hd):
    """General kernels: head_dim 48 (every CaiT size) and N > 256 (ViT-L/16 at 384^2: N = 577), online softmax; N > 608: the streaming
    kernels (same arithmetic, K / V and Q / dO through LDS in segments)."""
    rng = np.random.default_rng(B * 7 + N + hd)
    d = H * hd
    qkv = rb(rng.standard_normal((B * N, 3 * d)))
    qkv[:, :d] = rb(qkv[:, :d] / np.sqrt(hd) * 2.5)
    if N > 300:  # force a late running-max update: one key in the LAST chunk dominates query 3 (guide rule 26)
        qkv[3, :hd] = 3.0
        qkv[N - 2, d:d + hd] = 3.0
    d_o = rb(rng.standard_normal((B * N, d)))
    qkv_d = dev(qkv, bf16)
    o, lse = ops.attention_fwd(qkv_d, B, N, H, head_dim=hd)
    o_ref, lse_ref, _ = _attn_ref(qkv, B, N, H, hd)
    assert np.isfinite(host(o)).all()
    assert rel(host(o), o_ref) < 3e-3, rel(host(o), o_ref)
    assert np.abs(host(lse) - lse_ref).max() < 2e-4 * max(1.0, np.abs(lse_ref).max())
    # with the kernels' operand roundings emulated (online softmax: 128-key steps when the general kernel runs)
    general = hd != 64 or N > 256
    o_em, g_em = _attn_emul(qkv, B, N, H, hd, d_o=d_o, o_saved=host(o), chunk=128 if general else None)
    e = rel(host(o), rb(o_em))
    assert e < 6e-4, ("o vs emulated", e)
    t = torch.tensor(qkv.astype(np.float64), requires_grad=True)
    x = t.view(B, N, 3, H, hd)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * N, d).backward(torch.tensor(d_o.astype(np.float64)))
    g = t.grad.numpy()
    dqkv = torch.full((B * N, 3 * d), float("nan"), dtype=bf16, device="cuda")
    ops.attention_bwd(qkv_d, o, dev(d_o, bf16), lse, B, N, H, dq_scale=1.0, dqkv=dqkv, head_dim=hd)
    out = host(dqkv)
    assert np.isfinite(out).all()
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        r = rel(out[:, sl], g[:, sl])
        assert r < 1.2e-2, (name, r)  # vs exact autograd (no operand rounding, exact O)
        r = rel(out[:, sl], rb(g_em[:, sl]))
        assert r < 1e-3, (name, "vs emulated", r)


# (37, 197, 12), (64, 197, 12), (65, 64, 12): more (image, head) items than resident workgroups - the persistent kernel's item loop with
# two and three items per workgroup; N = 225, 256: the one-item kernel (the persistent form's LDS does not fit above N = 224)
@pytest.mark.parametrize("B,N,H", [(2, 197, 3), (2, 196, 2), (1, 50, 4), (1, 33, 1), (1, 256, 1), (3, 224, 2), (2, 225, 1), (37, 197, 12),
                                   (64, 197, 12), (65, 64, 12), (5, 97, 3), (3, 129, 2), (2, 161, 5), (1, 1, 1)])
def test_attention_bwd(ops, B, N, H):
    rng = np.random.default_rng(B * 10 + N)
    d = H * 64
    qkv = rb(rng.standard_normal((B * N, 3 * d)))
    qkv[:, :d] = rb(qkv[:, :d] / 8.0 * 2.0)
    d_o = rb(rng.standard_normal((B * N, d)))
    t = torch.tensor(qkv.astype(np.float64), requires_grad=True)
    x = t.view(B, N, 3, H, 64)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    o_t = (p @ v).permute(0, 2, 1, 3).reshape(B * N, d)
    o_t.backward(torch.tensor(d_o.astype(np.float64)))
    g = t.grad.numpy()
    qkv_d = dev(qkv, bf16)
    o, lse = ops.attention_fwd(qkv_d, B, N, H)
    dqkv = ops.attention_bwd(qkv_d, o, dev(d_o, bf16), lse, B, N, H, dq_scale=1.0)
    out = host(dqkv)
    assert np.isfinite(out).all()
    _, g_em = _attn_emul(qkv, B, N, H, 64, d_o=d_o, o_saved=host(o))
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        r = rel(out[:, sl], g[:, sl])
        assert r < 1e-2, (name, r)  # vs exact autograd: P, dS pass through bf16 MFMA operands; O is bf16
        r = rel(out[:, sl], rb(g_em[:, sl]))
        assert r < 1e-3, (name, "vs emulated", r)  # same roundings emulated: summation order + output-rounding flips remain
    dq2 = host(ops.attention_bwd(qkv_d, o, dev(d_o, bf16), lse, B, N, H, dq_scale=0.125))[:, :d]
    assert rel(dq2, g[:, :d] * 0.125) < 1e-2
    # the dQ shares of the key-owning waves meet in LDS in a fixed order: repeated launches agree bit for bit
    for _ in range(3):
        assert torch.equal(ops.attention_bwd(qkv_d, o, dev(d_o, bf16), lse, B, N, H, dq_scale=1.0), dqkv)


# ------------------------------------------------------------------------------------------ train ops
def test_cls_pos_and_grad(ops):
    rng = np.random.default_rng(0)
    B, N, d = 5, 17, 64
    cls = rng.standard_normal(d).astype(np.float32)
    pos = rng.standard_normal((N, d)).astype(np.float32)
    x0 = torch.zeros((B * N, d), device="cuda")
    ops.cls_pos_rows(dev(cls), dev(pos), x0, B, N)
    out = host(x0).reshape(B, N, d)
    assert np.allclose(out[:, 0], cls + pos[0]) and np.all(out[:, 1:] == 0)
    dx0 = rng.standard_normal((B * N, d)).astype(np.float32)
    dpos = torch.zeros((N, d), device="cuda")
    dcls = torch.zeros(d, device="cuda")
    ops.pos_cls_grad(dev(dx0), dpos, dcls, B, N)
    assert rel(host(dpos), dx0.reshape(B, N, d).sum(0)) < 1e-6
    assert rel(host(dcls), dx0.reshape(B, N, d)[:, 0].sum(0)) < 1e-6


@pytest.mark.parametrize("mix", [False, True])
def test_softmax_xent(ops, mix):
    rng = np.random.default_rng(1)
    B, C, Cp = 37, 1000, 1024
    logits = (rng.standard_normal((B, C)) * 3).astype(np.float32)
    labels = rng.integers(0, C, B)
    l2 = rng.integers(0, C, B) if mix else None
    ratio = rng.random(B).astype(np.float32) if mix else None
    loss_rows = torch.zeros(B, device="cuda")
    loss_mean = torch.zeros(1, device="cuda")
    dz = torch.full((B, Cp), 3.0, dtype=bf16, device="cuda")
    dbias = torch.zeros(C, device="cuda")
    t1 = torch.zeros(B, device="cuda")
    t5 = torch.zeros(B, device="cuda")
    ops.softmax_xent(dev(logits), dev(labels.astype(np.int32)), 0.1, None, None if l2 is None else dev(l2.astype(np.int32)),
                     None if ratio is None else dev(ratio), loss_rows, loss_mean, dz, dbias, t1, t5)
    ref = vit_ref.loss_fn(logits.astype(np.float64), labels, 0.1, l2, ratio)
    assert abs(float(loss_mean) - ref) < 1e-5 * max(1.0, abs(ref))
    if not mix:
        g = vit_ref.dloss_dlogits(logits.astype(np.float64), labels, 0.1)
        assert rel(host(dz)[:, :C], rb(g)) < 1e-3
        assert float(dz[:, C:].abs().max()) == 0
        assert rel(host(dbias), host(dz)[:, :C].astype(np.float64).sum(0)) < 1e-4
        tk = vit_ref.topk_correct(logits, labels)
        assert np.array_equal(host(t1), tk["top_1_acc"]) and np.array_equal(host(t5), tk["top_5_acc"])


def test_adamw_and_clip(ops):
    rng = np.random.default_rng(2)
    n = 4096 + 4
    p0 = rng.standard_normal(n).astype(np.float32)
    p, m, v = p0.astype(np.float64), np.zeros(n), np.zeros(n)
    pt, mt, vt = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32) * 3
        ss = torch.zeros(1, device="cuda")
        ops.sumsq(dev(g), ss)
        norm = np.sqrt((g.astype(np.float64) ** 2).sum())
        assert abs(float(ss) - norm ** 2) < 1e-3 * norm ** 2
        clip = 1.0 if norm < 1.0 else 1.0 / norm
        p, m, v = vit_ref.adamw_update(p, g.astype(np.float64), m, v, step, 3e-3, 1e-4, clip)
        ops.adamw_step(pt, dev(g), mt, vt, step, 3e-3, weight_decay=1e-4, grad_sumsq=ss, max_norm=1.0)
    assert rel(host(pt), p) < 1e-6 and rel(host(mt), m) < 1e-5 and rel(host(vt), v) < 1e-4  # fp32 moments vs fp64 oracle


@pytest.mark.parametrize("n", [8 * 4096, 8 * 4096 + 4, 4])
def test_adamw_mirror_is_bf16_of_the_updated_parameters(ops, n):
    """savit_adamw_step_mirror: the same update as savit_adamw_step (bitwise) plus bf16(params) in the same flat layout - the [in, out]
    MFMA operands of the input-gradient GEMMs are views into that mirror (train.py:25-27,100 + the bf16 cast of train.py:81's graph)."""
    g0 = torch.Generator(device="cuda").manual_seed(n)
    p = torch.randn(n, device="cuda", generator=g0)
    g = torch.randn(n, device="cuda", generator=g0) * 2
    a = [p.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
    b = [p.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
    mirror = torch.full((n,), float("nan"), dtype=bf16, device="cuda")
    ss = torch.zeros(1, device="cuda")
    ops.sumsq(g, ss)
    for step in (1, 2):
        ops.adamw_step(a[0], g, a[1], a[2], step, 1e-2, weight_decay=1e-3, grad_sumsq=ss, max_norm=1.0)
        ops.adamw_step(b[0], g, b[1], b[2], step, 1e-2, weight_decay=1e-3, grad_sumsq=ss, max_norm=1.0, mirror=mirror)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(mirror, b[0].to(bf16))
    assert not torch.equal(b[0], p)


def test_layernorm_bwd_with_the_bias_gradient_slab_riding_along(ops):
    """savit_layernorm_bwd_ex = savit_layernorm_bwd + savit_colsum_finalize(accumulate) in its finalize launch: same dx, same LN
    gradients, and the slab's column sums added to extra_out in a fixed order (bitwise repeatable)."""
    rng = np.random.default_rng(21)
    rows, d, nx, rx = 1000, 192, 768, 158
    x = dev(rng.standard_normal((rows, d)).astype(np.float32) * 2 + 0.5)
    gam = dev((1 + 0.1 * rng.standard_normal(d)).astype(np.float32))
    bet = dev((0.1 * rng.standard_normal(d)).astype(np.float32))
    y, mean, rstd = ops.layernorm_fwd(x, gam, bet)
    dy = dev(rng.standard_normal((rows, d)).astype(np.float32), bf16)
    slab = dev(rng.standard_normal((rx, nx)).astype(np.float32))
    base = dev(rng.standard_normal(nx).astype(np.float32))

    def run(extra):
        dg, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
        out = base.clone()
        dx = ops.layernorm_bwd(dy, x, gam, mean, rstd, dg, db, **(dict(extra_slab=slab, extra_out=out) if extra else {}))
        return dx, dg, db, out

    dx0, dg0, db0, _ = run(False)
    dx1, dg1, db1, out1 = run(True)
    assert torch.equal(dx0, dx1)
    assert rel(host(dg1), host(dg0)) < 1e-6 and rel(host(db1), host(db0)) < 1e-6  # (a few fp32 atomic adders finish these)
    ref = host(base).astype(np.float64) + host(slab).astype(np.float64).sum(0)
    assert rel(host(out1), ref) < 1e-6
    assert torch.equal(out1, run(True)[3])
    with pytest.raises(ValueError):
        ops.layernorm_bwd(dy, x, gam, mean, rstd, dg0, db0, extra_slab=slab, extra_out=base[:-4])


def test_cast_transpose_and_layout(ops):
    rng = np.random.default_rng(3)
    L, R, C, bs = 3, 100, 72, 100 * 72 + 40
    src = rng.standard_normal(L * bs).astype(np.float32)
    dn = torch.zeros(L * R * C, dtype=bf16, device="cuda")
    dt = torch.zeros(L * R * C, dtype=bf16, device="cuda")
    ops.cast_transpose_bf16(dev(src), L, R, C, bs, dn, R * C, dt, R * C)
    for l in range(L):
        mat = src[l * bs:l * bs + R * C].reshape(R, C)
        assert np.array_equal(host(dn)[l * R * C:(l + 1) * R * C].reshape(R, C), rb(mat))
        assert np.array_equal(host(dt)[l * R * C:(l + 1) * R * C].reshape(C, R), rb(mat.T))
    x = rng.standard_normal((8, 6, 3, 5)).astype(np.float32)  # H W C N
    y = ops.hwcn_to_nhwc_bf16(dev(x))
    assert np.array_equal(host(y), rb(np.transpose(x, (3, 0, 1, 2))))
    z = rng.standard_normal(1003).astype(np.float32)
    assert np.array_equal(host(ops.cast_bf16(dev(z), torch.empty(1003, dtype=bf16, device="cuda"))), rb(z))


def _th_emul(qkv, T1, T2, B, N, H, hd, d_o=None, dq_scale=1.0):
    """fp64 math WITH the bf16 roundings of the talking-heads kernels (attention.hip th_*, th_fused.hip): S = bf16(Q K^T) and
    P' = bf16(mix_T2(softmax(mix_T1(S)))) are the stored / MFMA-operand tensors of forward; backward rounds dP' = bf16(dO V^T) and
    dS = bf16(mix_T1^T(P (dP - sum_k P dP))) and takes dT2 = sum P x dP' (fp32 P, bf16 dP'), dT1 = sum S x dS_c (bf16 S, fp32 dS_c).
    Returns o (and dqkv, dT1, dT2), unrounded."""
    d = H * hd
    x = qkv.astype(np.float64).reshape(B, N, 3, H, hd)
    q, k, v = (x[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    T1, T2 = T1.astype(np.float64), T2.astype(np.float64)
    S = rb(q @ k.transpose(0, 1, 3, 2)).astype(np.float64)
    sc = np.einsum("hi,bhqk->biqk", T1, S)
    e = np.exp(sc - sc.max(-1, keepdims=True))
    P = e / e.sum(-1, keepdims=True)
    Pp = rb(np.einsum("hi,bhqk->biqk", T2, P)).astype(np.float64)
    o = (Pp @ v).transpose(0, 2, 1, 3).reshape(B * N, d)
    if d_o is None:
        return o
    do = d_o.astype(np.float64).reshape(B, N, H, hd).transpose(0, 2, 1, 3)
    dPp = rb(do @ v.transpose(0, 1, 3, 2)).astype(np.float64)
    dT2 = np.einsum("bhqk,biqk->hi", P, dPp)
    dP = np.einsum("hi,biqk->bhqk", T2, dPp)
    dsc = P * (dP - (P * dP).sum(-1, keepdims=True))
    dT1 = np.einsum("bhqk,biqk->hi", S, dsc)
    dS = rb(np.einsum("hi,biqk->bhqk", T1, dsc)).astype(np.float64)
    dq, dk, dv = (dS @ k) * dq_scale, dS.transpose(0, 1, 3, 2) @ q, Pp.transpose(0, 1, 3, 2) @ do
    g = np.stack([t.transpose(0, 2, 1, 3) for t in (dq, dk, dv)], axis=2).reshape(B * N, 3 * d)
    return o, g, dT1, dT2


# measured (round 3, both the materialising and the fused kernels, 14 geometries): o <= 2.0e-4, dq / dk / dv <= 1.7e-4, dT1 / dT2 <= 4.1e-5
TH_EMUL_BARS = {"o": 6e-4, "dq": 6e-4, "dk": 6e-4, "dv": 6e-4, "dT1": 2e-4, "dT2": 2e-4}


# ------------------------------------------------------------------------------------------ talking-heads attention (CaiT)
@pytest.mark.parametrize("B,N,H,hd", [(2, 196, 8, 48), (1, 196, 4, 48), (2, 50, 6, 48), (1, 33, 2, 64), (1, 197, 8, 48),
                                      (2, 196, 16, 48), (1, 37, 16, 48)])  # 16 heads: cait_m_* (dT reduced as four 8x8 tiles)
def test_talking_heads_attention(ops, B, N, H, hd):
    """attention.py:41-58 with talking_heads=True: fp64 autograd reference incl. dT1/dT2 (talking_heads.py:13)."""
    rng = np.random.default_rng(B + N + H)
    d = H * hd
    qkv = rb(rng.standard_normal((B * N, 3 * d)))
    qkv[:, :d] = rb(qkv[:, :d] / np.sqrt(hd) * 2.0)
    T1 = (np.linalg.qr(rng.standard_normal((H, H)))[0] + 0.1 * rng.standard_normal((H, H))).astype(np.float32)
    T2 = (np.linalg.qr(rng.standard_normal((H, H)))[0] + 0.1 * rng.standard_normal((H, H))).astype(np.float32)
    d_o = rb(rng.standard_normal((B * N, d)))
    t = torch.tensor(qkv.astype(np.float64), requires_grad=True)
    t1 = torch.tensor(T1.astype(np.float64), requires_grad=True)
    t2 = torch.tensor(T2.astype(np.float64), requires_grad=True)
    x = t.view(B, N, 3, H, hd)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    s_raw = q @ k.transpose(-1, -2)
    # the reference's bf16 graph materialises S in bf16 before the fp32 talking-heads mix: emulate that rounding
    # (straight-through for the gradient), as the kernel stores S as bf16
    s_rnd = s_raw + (s_raw.detach().float().bfloat16().double() - s_raw.detach())
    sc = torch.einsum("hi,bhqk->biqk", t1, s_rnd)
    w = torch.einsum("hi,bhqk->biqk", t2, torch.softmax(sc, dim=-1))
    o_t = (w @ v).permute(0, 2, 1, 3).reshape(B * N, d)
    o_t.backward(torch.tensor(d_o.astype(np.float64)))
    qkv_d, T1d, T2d = dev(qkv, bf16), dev(T1), dev(T2)
    o, s_buf, p_buf = ops.th_attention_fwd(qkv_d, T1d, T2d, B, N, H, head_dim=hd)
    assert np.isfinite(host(o)).all()
    assert rel(host(o), o_t.detach().numpy()) < 5e-3, rel(host(o), o_t.detach().numpy())  # P' is a bf16 MFMA operand
    dT1 = torch.zeros((H, H), device="cuda")
    dT2 = torch.zeros((H, H), device="cuda")
    dqkv = ops.th_attention_bwd(qkv_d, T1d, T2d, s_buf, p_buf, dev(d_o, bf16), dT1, dT2, B, N, H, dq_scale=1.0, head_dim=hd)
    out, g = host(dqkv), t.grad.numpy()
    assert np.isfinite(out).all()
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        assert rel(out[:, sl], g[:, sl]) < 2e-2, (name, rel(out[:, sl], g[:, sl]))
    assert rel(host(dT1), t1.grad.numpy()) < 2e-2, rel(host(dT1), t1.grad.numpy())
    assert rel(host(dT2), t2.grad.numpy()) < 2e-2, rel(host(dT2), t2.grad.numpy())
    # the same roundings emulated (bf16 S, P', dP', dS): what remains is summation order and flips of the bf16 outputs / stored tensors
    o_em, g_em, dT1_em, dT2_em = _th_emul(qkv, T1, T2, B, N, H, hd, d_o=d_o)
    worst = {"o": rel(host(o), rb(o_em)), "dT1": rel(host(dT1), dT1_em), "dT2": rel(host(dT2), dT2_em)}
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        worst[name] = rel(out[:, sl], rb(g_em[:, sl]))
    print(f"[th emul {B},{N},{H},{hd}] " + " ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    for k_, v_ in worst.items():
        assert v_ < TH_EMUL_BARS[k_], (k_, v_)


@pytest.mark.parametrize("B,N,H,hd", [(2, 196, 8, 48), (1, 196, 4, 48), (2, 50, 6, 48), (1, 33, 2, 64), (1, 197, 8, 48), (3, 208, 8, 64), (2, 17, 8, 48)])
def test_fused_talking_heads_attention(ops, B, N, H, hd):
    """csrc/th_fused.hip (S / P' in LDS) against the same fp64 autograd reference and tolerances as the materialising kernels, and
    against those kernels themselves (same roundings: bf16 S, P', dP', dS)."""
    assert ops.th_fused_supported(N, H, hd)
    rng = np.random.default_rng(B + N + H)
    d = H * hd
    qkv = rb(rng.standard_normal((B * N, 3 * d)))
    qkv[:, :d] = rb(qkv[:, :d] / np.sqrt(hd) * 2.0)
    T1 = (np.linalg.qr(rng.standard_normal((H, H)))[0] + 0.1 * rng.standard_normal((H, H))).astype(np.float32)
    T2 = (np.linalg.qr(rng.standard_normal((H, H)))[0] + 0.1 * rng.standard_normal((H, H))).astype(np.float32)
    d_o = rb(rng.standard_normal((B * N, d)))
    t = torch.tensor(qkv.astype(np.float64), requires_grad=True)
    t1 = torch.tensor(T1.astype(np.float64), requires_grad=True)
    t2 = torch.tensor(T2.astype(np.float64), requires_grad=True)
    x = t.view(B, N, 3, H, hd)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    s_raw = q @ k.transpose(-1, -2)
    s_rnd = s_raw + (s_raw.detach().float().bfloat16().double() - s_raw.detach())
    sc = torch.einsum("hi,bhqk->biqk", t1, s_rnd)
    w = torch.einsum("hi,bhqk->biqk", t2, torch.softmax(sc, dim=-1))
    o_t = (w @ v).permute(0, 2, 1, 3).reshape(B * N, d)
    o_t.backward(torch.tensor(d_o.astype(np.float64)))
    qkv_d, T1d, T2d = dev(qkv, bf16), dev(T1), dev(T2)
    o = ops.th_fused_attention_fwd(qkv_d, T1d, T2d, B, N, H, head_dim=hd)
    assert np.isfinite(host(o)).all()
    assert rel(host(o), o_t.detach().numpy()) < 5e-3, rel(host(o), o_t.detach().numpy())
    dT1 = torch.zeros((H, H), device="cuda")
    dT2 = torch.zeros((H, H), device="cuda")
    dqkv = ops.th_fused_attention_bwd(qkv_d, T1d, T2d, dev(d_o, bf16), dT1, dT2, B, N, H, dq_scale=1.0, head_dim=hd)
    out, g = host(dqkv), t.grad.numpy()
    assert np.isfinite(out).all()
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        assert rel(out[:, sl], g[:, sl]) < 2e-2, (name, rel(out[:, sl], g[:, sl]))
    assert rel(host(dT1), t1.grad.numpy()) < 2e-2, rel(host(dT1), t1.grad.numpy())
    assert rel(host(dT2), t2.grad.numpy()) < 2e-2, rel(host(dT2), t2.grad.numpy())
    o_em, g_em, dT1_em, dT2_em = _th_emul(qkv, T1, T2, B, N, H, hd, d_o=d_o)  # same roundings emulated in fp64
    worst = {"o": rel(host(o), rb(o_em)), "dT1": rel(host(dT1), dT1_em), "dT2": rel(host(dT2), dT2_em)}
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        worst[name] = rel(out[:, sl], rb(g_em[:, sl]))
    print(f"[th fused emul {B},{N},{H},{hd}] " + " ".join(f"{k} {v:.1e}" for k, v in worst.items()))
    for k_, v_ in worst.items():
        assert v_ < TH_EMUL_BARS[k_], (k_, v_)
    # the materialising path computes the same function with the same roundings (different MFMA shapes: not bitwise)
    o_m, s_buf, p_buf = ops.th_attention_fwd(qkv_d, T1d, T2d, B, N, H, head_dim=hd)
    assert rel(host(o), host(o_m)) < 2e-3, rel(host(o), host(o_m))
    dT1m = torch.zeros((H, H), device="cuda")
    dT2m = torch.zeros((H, H), device="cuda")
    dqkv_m = ops.th_attention_bwd(qkv_d, T1d, T2d, s_buf, p_buf, dev(d_o, bf16), dT1m, dT2m, B, N, H, dq_scale=1.0, head_dim=hd)
    assert rel(out, host(dqkv_m)) < 4e-3, rel(out, host(dqkv_m))
    assert rel(host(dT1), host(dT1m)) < 2e-3 and rel(host(dT2), host(dT2m)) < 2e-3
    # accumulation into dT1 / dT2 and run-to-run determinism
    dqkv2 = ops.th_fused_attention_bwd(qkv_d, T1d, T2d, dev(d_o, bf16), dT1, dT2, B, N, H, dq_scale=1.0, head_dim=hd)
    assert torch.equal(dqkv2, dqkv)
    assert rel(host(dT1), 2 * t1.grad.numpy()) < 2e-2


def test_fused_talking_heads_geometry_gate(ops):
    assert not ops.th_fused_supported(196, 16, 48) and not ops.th_fused_supported(209, 8, 48) and not ops.th_fused_supported(196, 8, 32)
    with pytest.raises(ValueError, match="not covered"):
        ops.th_fused_attention_fwd(torch.zeros(2 * 196, 3 * 16 * 48, dtype=bf16, device="cuda"), torch.eye(16, device="cuda"),
                                   torch.eye(16, device="cuda"), 2, 196, 16, head_dim=48)


@pytest.mark.parametrize("rows,d,rps,with_bias,with_rs", [(2 * 196, 384, 196, True, True), (5 * 197, 192, 197, False, True), (7, 768, 1, True, False),
                                                          (3 * 50, 128, 50, True, True), (64, 1024, 1, False, False)])
def test_layernorm_bwd_ls_equals_the_two_launches(ops, rows, d, rps, with_bias, with_rs):
    """savit_layernorm_bwd_ls = savit_layernorm_bwd followed by savit_layerscale_bwd on the residual gradient it wrote (the reverse of
    cait.py:28-60: LayerNorm VJP, then layerscale.py:18-23 / stochastic_depth.py:16-27 of the sub-block before it) - element by element
    the same arithmetic, so dx and dbranch must agree exactly and the column sums to summation order."""
    from savit_amd import lib as _lib

    L = _lib.load()
    rng = np.random.default_rng(rows * 3 + d)
    nsamp = rows // rps
    x = (rng.standard_normal((rows, d)) * 2 + 0.3).astype(np.float32)
    dy = rb(rng.standard_normal((rows, d)))
    dres = rng.standard_normal((rows, d)).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(d)).astype(np.float32)
    branch = rb(rng.standard_normal((rows, d)))
    ls = (0.5 + rng.random(d)).astype(np.float32)
    rs = np.where(rng.random(nsamp) < 0.7, 1.0 / 0.9, 0.0).astype(np.float32)
    mean = x.mean(1).astype(np.float32)
    rstd = (1.0 / np.sqrt(x.var(1) + 1e-6)).astype(np.float32)
    st = torch.cuda.current_stream().cuda_stream
    ws = int(L.savit_layernorm_bwd_workspace_bytes(rows, d))
    t_ws = torch.empty(max(ws, 16), dtype=torch.uint8, device="cuda")
    t_x, t_dy, t_g, t_m, t_r = dev(x), dev(dy, bf16), dev(gamma), dev(mean), dev(rstd)
    t_br, t_ls, t_rs = dev(branch, bf16), dev(ls), dev(rs)
    init = {k: rng.standard_normal(d).astype(np.float32) for k in ("dg", "db", "dls", "dbias")}  # the column sums ACCUMULATE

    xs, xo0 = rng.standard_normal((37, 64)).astype(np.float32), rng.standard_normal(64).astype(np.float32)
    t_xs, t_xo = dev(xs), dev(xo0.copy())

    def run(fused):
        t_dres = dev(dres.copy())
        o = {k: dev(v.copy()) for k, v in init.items()}
        t_dbr = torch.full((rows, d), 7.0, dtype=bf16, device="cuda")
        dbias = o["dbias"].data_ptr() if with_bias else None
        rsp = t_rs.data_ptr() if with_rs else None
        if fused:
            rc = L.savit_layernorm_bwd_ls(t_dy.data_ptr(), t_x.data_ptr(), t_g.data_ptr(), t_m.data_ptr(), t_r.data_ptr(), t_dres.data_ptr(),
                                          t_dres.data_ptr(), o["dg"].data_ptr(), o["db"].data_ptr(), rows, d, d, d, 1, t_br.data_ptr(), t_ls.data_ptr(),
                                          rsp, rps, t_dbr.data_ptr(), o["dls"].data_ptr(), dbias, t_ws.data_ptr(), ws, t_xs.data_ptr(), 37, 64,
                                          t_xo.data_ptr(), st)
            assert rc == 0
            torch.cuda.synchronize()
            assert rel(host(t_xo), xo0 + xs.sum(0)) < 1e-6  # the extra slab's column sums ride along (savit_layernorm_bwd_ex's contract)
        else:
            rc = L.savit_layernorm_bwd(t_dy.data_ptr(), t_x.data_ptr(), t_g.data_ptr(), t_m.data_ptr(), t_r.data_ptr(), t_dres.data_ptr(),
                                       t_dres.data_ptr(), None, o["dg"].data_ptr(), o["db"].data_ptr(), None, rows, d, d, d, 1, t_ws.data_ptr(), ws, st)
            assert rc == 0
            rc = L.savit_layerscale_bwd(t_dres.data_ptr(), t_br.data_ptr(), t_ls.data_ptr(), rsp, rps, t_dbr.data_ptr(), o["dls"].data_ptr(), dbias,
                                        rows, d, d, t_ws.data_ptr(), ws, st)
            assert rc == 0
        torch.cuda.synchronize()
        return host(t_dres), host(t_dbr), {k: host(v) for k, v in o.items()}

    dx_f, dbr_f, o_f = run(True)
    dx_u, dbr_u, o_u = run(False)
    assert np.array_equal(dx_f, dx_u) and np.array_equal(dbr_f, dbr_u)
    for k in ("dg", "db", "dls") + (("dbias",) if with_bias else ()):
        assert rel(o_f[k] - init[k], o_u[k] - init[k]) < 1e-5, k
    if not with_bias:
        assert np.array_equal(o_f["dbias"], init["dbias"])
    # and against the formulas in fp64
    xh = (x.astype(np.float64) - mean[:, None]) * rstd[:, None]
    gy = dy.astype(np.float64) * rb(gamma)
    want_dx = rstd[:, None] * (gy - gy.mean(1, keepdims=True) - xh * (gy * xh).mean(1, keepdims=True)) + dres
    assert rel(dx_f, want_dx) < 1e-5
    rs_rows = (np.repeat(rs, rps)[:, None] if with_rs else np.ones((rows, 1))).astype(np.float64)
    assert rel(o_f["dls"] - init["dls"], (want_dx * rs_rows * branch).sum(0)) < 1e-4


# ------------------------------------------------------------------------------------------ LayerScale backward (row a9)
@pytest.mark.parametrize("rows,d,rps,with_bias", [(2 * 196, 384, 196, True), (5 * 197, 192, 197, False), (7, 768, 1, True), (64, 4096, 1, True)])
def test_layerscale_bwd(ops, rows, d, rps, with_bias):
    """out = res + rowscale[sample] * layerscale * branch (layerscale.py:18-23, stochastic_depth.py:16-27; cait.py:36-52):
    dbranch = bf16(dres * rs * ls), d_layerscale += sum_rows dres * rs * branch, dbias += column sums of dbranch."""
    import ctypes

    from savit_amd import lib as _lib

    L = _lib.load()
    rng = np.random.default_rng(rows + d)
    nsamp = rows // rps
    dres = rng.standard_normal((rows, d)).astype(np.float32)
    branch = rb(rng.standard_normal((rows, d)))
    ls = (0.5 + rng.random(d)).astype(np.float32)
    rs = np.where(rng.random(nsamp) < 0.7, 1.0 / 0.9, 0.0).astype(np.float32)  # stochastic-depth keep masks / keep (dropped samples: 0)
    dls0 = rng.standard_normal(d).astype(np.float32)  # the outputs ACCUMULATE
    db0 = rng.standard_normal(d).astype(np.float32)
    t_dres, t_br, t_ls, t_rs = dev(dres), dev(branch, bf16), dev(ls), dev(rs)
    t_dbr = torch.empty(rows, d, dtype=bf16, device="cuda")
    t_dls, t_db = dev(dls0.copy()), dev(db0.copy())
    ws = int(L.savit_layernorm_bwd_workspace_bytes(rows, d))
    t_ws = torch.empty(max(ws, 16), dtype=torch.uint8, device="cuda")
    rc = L.savit_layerscale_bwd(t_dres.data_ptr(), t_br.data_ptr(), t_ls.data_ptr(), t_rs.data_ptr(), rps, t_dbr.data_ptr(), t_dls.data_ptr(),
                                t_db.data_ptr() if with_bias else None, rows, d, d, t_ws.data_ptr(), ws, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    rs_rows = np.repeat(rs, rps)[:, None].astype(np.float64)
    want_dbr = dres.astype(np.float64) * rs_rows * ls
    got_dbr = host(t_dbr)
    assert np.array_equal(got_dbr, rb(want_dbr.astype(np.float32))) or rel(got_dbr, rb(want_dbr.astype(np.float32))) < 1e-3
    want_dls = dls0 + (dres.astype(np.float64) * rs_rows * branch).sum(0)
    assert rel(host(t_dls), want_dls) < 2e-5, rel(host(t_dls), want_dls)
    if with_bias:
        want_db = db0 + rb(want_dbr.astype(np.float32)).astype(np.float64).sum(0)
        assert rel(host(t_db), want_db) < 2e-5, rel(host(t_db), want_db)
    assert np.all(got_dbr[np.repeat(rs, rps) == 0] == 0)  # a dropped sample sends no cotangent into its branch
    # vs autograd of the forward formula (fp64): same numbers by another route
    tb = torch.tensor(branch, dtype=torch.float64, requires_grad=True)
    tl = torch.tensor(ls, dtype=torch.float64, requires_grad=True)
    (torch.tensor(rs_rows) * tl * tb * torch.tensor(dres, dtype=torch.float64)).sum().backward()
    assert rel(host(t_dls) - dls0, tl.grad.numpy()) < 2e-4
    assert rel(got_dbr, tb.grad.numpy()) < 3e-3  # bf16 output


# ------------------------------------------------------------------------------------------ class attention (row a12)
@pytest.mark.parametrize("B,Nk,H,hd", [(3, 197, 8, 48), (2, 197, 4, 64), (5, 65, 6, 64), (2, 256, 16, 48)])
def test_class_attention_fwd_bwd(ops, B, Nk, H, hd):
    """ClassSelfAttentionBlock (cait.py:10-15 over attention.py:39-58): one (pre-scaled) query per image against Nk keys, packed the way
    the CaiT engine packs them - q at row b*Nk of a [B*Nk, 3d] buffer, k | v at columns d.. of every row.  Forward vs fp64 softmax
    attention on the same bf16 inputs (scores rounded to bf16 like the reference's bf16 score tensor), backward vs fp64 autograd."""
    import math

    from savit_amd import lib as _lib

    L = _lib.load()
    d = H * hd
    rng = np.random.default_rng(B * 1000 + Nk)
    qkv = rb(rng.standard_normal((B * Nk, 3 * d)) * 0.7)
    d_o = rb(rng.standard_normal((B, d)))
    t_qkv, t_do = dev(qkv, bf16), dev(d_o, bf16)
    t_o = torch.empty(B, d, dtype=bf16, device="cuda")
    t_p = torch.empty(B, H, Nk, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    qk = t_qkv.data_ptr()
    assert L.savit_class_attention_fwd(qk, Nk * 3 * d, qk + d * 2, 3 * d, t_o.data_ptr(), t_p.data_ptr(), B, Nk, H, hd, st) == 0
    t_dqkv = torch.zeros(B * Nk, 3 * d, dtype=bf16, device="cuda")
    dqk = t_dqkv.data_ptr()
    dqs = 1.0 / math.sqrt(hd)
    assert L.savit_class_attention_bwd(qk, Nk * 3 * d, qk + d * 2, 3 * d, t_p.data_ptr(), t_do.data_ptr(), dqk, Nk * 3 * d, dqk + d * 2, B, Nk,
                                       H, hd, dqs, st) == 0
    torch.cuda.synchronize()
    # fp64 reference; q0 is the UNSCALED query whose scaled, bf16-rounded value sits in the buffer: dq_scale maps d(scaled q) to d(q0)
    x = torch.tensor(qkv.reshape(B, Nk, 3, H, hd), dtype=torch.float64)
    q = x[:, 0, 0].clone().requires_grad_(True)   # [B,H,hd] (already scaled)
    k = x[:, :, 1].clone().requires_grad_(True)   # [B,Nk,H,hd]
    v = x[:, :, 2].clone().requires_grad_(True)
    s = torch.einsum("bhe,bkhe->bhk", q, k)
    s_b = s + (torch.tensor(rb(s.detach().numpy().astype(np.float32)), dtype=torch.float64) - s).detach()  # bf16 score tensor, straight-through
    p = torch.softmax(s_b, dim=-1)
    o = torch.einsum("bhk,bkhe->bhe", p, v)
    assert rel(host(t_p), p.detach().numpy()) < 1e-5
    o_want = o.detach().numpy().reshape(B, d)
    assert rel(host(t_o), o_want) < 3e-3, rel(host(t_o), o_want)
    assert np.abs(host(t_o) - o_want).max() <= 2.0 ** -7 * max(1.0, np.abs(o_want).max())
    o.backward(torch.tensor(d_o.reshape(B, H, hd), dtype=torch.float64))
    got = host(t_dqkv).reshape(B, Nk, 3, H, hd)
    assert rel(got[:, 0, 0], q.grad.numpy() * dqs) < 5e-3, rel(got[:, 0, 0], q.grad.numpy() * dqs)
    assert np.all(got[:, 1:, 0] == 0)  # only the cls row has a query
    assert rel(got[:, :, 1], k.grad.numpy()) < 5e-3, rel(got[:, :, 1], k.grad.numpy())
    assert rel(got[:, :, 2], v.grad.numpy()) < 5e-3, rel(got[:, :, 2], v.grad.numpy())
    # the kernel computes in fp32 on the bf16 inputs and rounds only its outputs: against the bf16-ROUNDED fp64 results what is left
    # is flips of the output rounding (fp32 vs fp64 just below / above a rounding boundary)
    ca = {"o": rel(host(t_o), rb(o_want)), "dq": rel(got[:, 0, 0], rb(q.grad.numpy() * dqs)), "dk": rel(got[:, :, 1], rb(k.grad.numpy())),
          "dv": rel(got[:, :, 2], rb(v.grad.numpy()))}
    print(f"[class attn emul {B},{Nk},{H},{hd}] " + " ".join(f"{k_} {v_:.1e}" for k_, v_ in ca.items()))
    for k_, v_ in ca.items():
        assert v_ < 1e-4, (k_, v_)  # measured <= 1.9e-5 (mostly bit-exact)


@pytest.mark.parametrize("B,N,H,hd", [(3, 197, 12, 64), (2, 577, 4, 64), (5, 50, 6, 64), (2, 197, 8, 48), (1, 640, 2, 64), (4, 1, 3, 64)])
def test_cls_query_attention_matches_the_dense_kernels_on_the_cls_row(ops, B, N, H, hd):
    """savit_cls_query_attention_fwd / _bwd (round 5: the last ViT layer's attention, whose cls output alone is read) against
    savit_attention_fwd / _bwd on the same packed qkv: output and LSE of query 0, and - with a cotangent that is zero off the cls rows -
    dQ of row 0, dK and dV of every row.  Same rounding points, other fp32 summation order: bf16 last-bit flips only."""
    from savit_amd import lib as _lib

    L = _lib.load()
    rng = np.random.default_rng(B * 1000 + N)
    d = H * hd
    qkv = rb(rng.standard_normal((B * N, 3 * d)))
    qkv[:, :d] = rb(qkv[:, :d] / np.sqrt(hd) * 2.0)
    qkv_d = dev(qkv, bf16)
    o, lse = ops.attention_fwd(qkv_d, B, N, H, head_dim=hd)
    oc = torch.empty((B, d), dtype=bf16, device="cuda")
    lc = torch.empty((B, H), dtype=torch.float32, device="cuda")
    s0 = torch.cuda.current_stream().cuda_stream
    rc = L.savit_cls_query_attention_fwd(qkv_d.data_ptr(), N * 3 * d, qkv_d.data_ptr() + 2 * d, 3 * d, oc.data_ptr(), lc.data_ptr(), B, N, H, hd, s0)
    assert rc == 0
    o_ref = host(o).reshape(B, N, d)[:, 0]
    assert rel(host(oc), o_ref) < 4e-3
    assert np.abs(host(lc) - host(lse).reshape(B, H, N)[:, :, 0]).max() < 1e-5 * max(1.0, np.abs(host(lse)).max())
    d_o = np.zeros((B, N, d), np.float32)
    d_o[:, 0] = rb(rng.standard_normal((B, d)))
    d_o_d = dev(d_o.reshape(B * N, d), bf16)
    ref = host(ops.attention_bwd(qkv_d, o, d_o_d, lse, B, N, H, dq_scale=0.5, head_dim=hd)).reshape(B, N, 3 * d)
    got = torch.full((B * N, 3 * d), 7.0, dtype=bf16, device="cuda")
    o0, lse0, do0 = o.view(B, N * d)[:, :d].contiguous(), lse.view(B, H, N)[:, :, 0].contiguous(), d_o_d.view(B, N * d)[:, :d].contiguous()
    rc = L.savit_cls_query_attention_bwd(qkv_d.data_ptr(), N * 3 * d, qkv_d.data_ptr() + 2 * d, 3 * d, o0.data_ptr(), lse0.data_ptr(), do0.data_ptr(),
                                         got.data_ptr(), N * 3 * d, got.data_ptr() + 2 * d, B, N, H, hd, 0.5, s0)
    assert rc == 0
    g = host(got).reshape(B, N, 3 * d)
    if N > 1:
        assert rel(g[:, 0, :d], ref[:, 0, :d]) < 4e-3                   # dQ of the cls query
    else:
        assert np.abs(g[:, 0, :d]).max() < 1e-5                           # one key: P = 1 and dS = 0 up to the rounding of O
    assert np.all(g[:, 1:, :d] == 7.0)                                    # other queries' rows are not written (the engine zero-fills once)
    assert rel(g[:, :, d:2 * d], ref[:, :, d:2 * d]) < 4e-3 or N == 1     # dK of every key
    assert rel(g[:, :, 2 * d:], ref[:, :, 2 * d:]) < 4e-3                 # dV of every key
    assert N == 1 or np.abs(ref[:, 1:, :d]).max() == 0.0                            # dense: zero cotangent rows give exact-zero dQ rows
    bad = L.savit_cls_query_attention_fwd(qkv_d.data_ptr(), N * 3 * d, qkv_d.data_ptr() + 2 * d, 3 * d, oc.data_ptr(), lc.data_ptr(), B, 641, H, hd, s0)
    assert bad == _lib.SAVIT_EINVAL


def test_attention_cu_budget_changes_the_grid_not_the_result(ops):
    """savit_set_cu_budget: the persistent attention kernels walk the same items with fewer workgroups - bit-identical outputs."""
    from savit_amd import lib as _lib

    L = _lib.load()
    B, N, H = 40, 197, 12
    d = H * 64
    rng = np.random.default_rng(5)
    qkv = dev(rb(rng.standard_normal((B * N, 3 * d)) * 0.5), bf16)
    d_o = dev(rb(rng.standard_normal((B * N, d))), bf16)
    outs = []
    try:
        for budget in (0, 240, 100, 7):
            assert L.savit_set_cu_budget(budget) == 0
            o, lse = ops.attention_fwd(qkv, B, N, H)
            outs.append((o, lse, ops.attention_bwd(qkv, o, d_o, lse, B, N, H, dq_scale=0.125)))
    finally:
        L.savit_set_cu_budget(0)
    assert L.savit_set_cu_budget(-1) != 0
    for o, lse, dq in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(lse, outs[0][1]) and torch.equal(dq, outs[0][2])


# ------------------------------------------------------------------------------------------ round 5: first touch, ranges, deferred column sums
@pytest.mark.parametrize("tile", [256, 640, 128])
def test_wgrad_grouped_first_touch_and_sumsq(ops, tile):
    """savit_wgrad_problem.overwrite: dW = X^T dY whatever dW held (NaNs here: a first-touch launch must not read it), bitwise the
    accumulate form on a zeroed dW; a weight cut over two launches; and the 32 accumulators of savit_gemm_bf16_wgrad_grouped_ex hold
    the sum of squares of everything stored (the weight gradients' share of the global norm, train.py:25)."""
    rng = np.random.default_rng(50 + tile)
    shapes = [(1999, 768, 1024), (1999, 256, 256), (3001, 384, 1152), (517, 384, 1000), (64, 128, 128)]
    probs = []
    for M, Kin, Nout in shapes:
        probs.append((dev(_mk(rng, M, Kin), bf16), dev(_mk(rng, M, Nout), bf16)))
    acc = [(X, dY, torch.zeros((X.shape[1], dY.shape[1]), dtype=torch.float32, device="cuda")) for X, dY in probs]
    ops.gemm_wgrad_grouped(acc, tile=tile)
    slots = torch.zeros(32, dtype=torch.float32, device="cuda")
    ft = [(X, dY, torch.full((X.shape[1], dY.shape[1]), float("nan"), dtype=torch.float32, device="cuda")) for X, dY in probs]
    ops.gemm_wgrad_grouped(ft, tile=tile, overwrite=True, sumsq32=slots)
    want = 0.0
    for (_, _, a), (_, _, b) in zip(acc, ft):
        assert torch.equal(a, b)
        want += float((a.double() ** 2).sum())
    assert abs(float(slots.double().sum()) - want) < 2e-6 * want
    assert int((slots != 0).sum()) >= 8  # spread over the accumulators, not one address
    # a weight cut over two launches, both first touch
    X, dY = probs[0]
    nt = ops._lib.load().savit_gemm_wgrad_group_tiles(768, 1024, tile)
    halves = torch.full((768, 1024), float("nan"), dtype=torch.float32, device="cuda")
    ops.gemm_wgrad_grouped([(X, dY, halves, 0, 5)], tile=tile, overwrite=True)
    ops.gemm_wgrad_grouped([(X, dY, halves, 5, nt - 5)], tile=tile, overwrite=True)
    assert torch.equal(halves, acc[0][2])
    # accumulate mode with the accumulators: what is stored (old + tile) is what is squared
    slots.zero_()
    ops.gemm_wgrad_grouped(acc, tile=tile, sumsq32=slots)
    want2 = sum(float((a.double() ** 2).sum()) for _, _, a in acc)
    assert abs(float(slots.double().sum()) - want2) < 2e-6 * want2 and abs(want2 - 4 * want) < 1e-5 * want2


def test_zero_and_sumsq_ranges(ops):
    """savit_zero_ranges / savit_sumsq_ranges: the passes over what the first-touch weight gradients leave (biases, LayerNorm
    parameters, embeddings): exactly the listed ranges are cleared / summed, nothing else is touched; more ranges than one launch takes."""
    n = 3_000_000
    g = torch.randn(n, device="cuda")
    keep = g.clone()
    ranges = [(0, 4), (1024, 768), (5000, 2048), (7048, 4096 + 4), (100_000, 590_000), (2_999_996, 4)] + [(1_000_000 + 64 * i, 32) for i in range(200)]
    mask = torch.zeros(n, dtype=torch.bool, device="cuda")
    for o, c in ranges:
        mask[o:o + c] = True
    out = torch.zeros(1, device="cuda")
    slots = torch.arange(32, device="cuda", dtype=torch.float32)
    ops.sumsq_ranges(g, ranges, out, slots)
    want = float((g[mask].double() ** 2).sum()) + float(slots.sum())
    assert abs(float(out) - want) < 1e-5 * want
    ops.zero_ranges(g, ranges)
    assert bool((g[mask] == 0).all()) and torch.equal(g[~mask], keep[~mask])
    with pytest.raises(ValueError):
        ops.zero_ranges(g, [(2, 4)])
    out.zero_()
    ops.sumsq_ranges(g, [], out, slots)  # no ranges: the accumulators alone
    assert abs(float(out) - float(slots.sum())) < 1e-3


@pytest.mark.parametrize("rows,d", [(197 * 8, 768), (577 * 3, 1024), (394, 384)])
def test_layernorm_bwd_deferred_finalize_jobs(ops, rows, d):
    """Deferred column sums: three layernorm_bwd calls with every output pointer None leave their slabs in workspaces of their own;
    ONE savit_layernorm_bwd_finalize_jobs launch then yields dgamma / dbeta / dcolsum / the extra slab's column sums of all three -
    the same values as the self-finalizing calls (same per-block partials; the order of the 8 atomic adds per column is the only
    freedom), and the same dx bit for bit."""
    rng = np.random.default_rng(rows + d)
    L = ops._lib.load()
    calls, jobs = [], []
    for k in range(3):
        x = dev((rng.standard_normal((rows, d)) * 1.5).astype(np.float32))
        dy = dev(_mk(rng, rows, d), bf16)
        gamma = dev((1 + 0.1 * rng.standard_normal(d)).astype(np.float32))
        _, mean, rstd = ops.layernorm_fwd(x, gamma, torch.zeros_like(gamma))
        dres = dev(_mk(rng, rows, d))
        slab = dev(_mk(rng, 37 + k, 4 * d)) if k != 1 else None
        calls.append((dy, x, gamma, mean, rstd, dres, slab))
    ref = []
    for dy, x, gamma, mean, rstd, dres, slab in calls:
        dg, db, dc = (torch.zeros(d, device="cuda") for _ in range(3))
        xo = torch.zeros(4 * d, device="cuda")
        dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, dres_in=dres, dcolsum=dc, **({"extra_slab": slab, "extra_out": xo} if slab is not None else {}))
        ref.append((dx, dg, db, dc, xo))
    outs = []
    for dy, x, gamma, mean, rstd, dres, slab in calls:
        ws = torch.empty(int(L.savit_layernorm_bwd_workspace_bytes(rows, d)), dtype=torch.uint8, device="cuda")
        dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, None, None, dres_in=dres, workspace=ws)
        dg, db, dc = (torch.zeros(d, device="cuda") for _ in range(3))
        xo = torch.zeros(4 * d, device="cuda")
        jobs.append((ws, rows, d, 3, (dg, db, dc, None), (slab, xo) if slab is not None else None))
        outs.append((dx, dg, db, dc, xo))
    ops.layernorm_bwd_finalize_jobs(jobs)
    for (dx0, dg0, db0, dc0, xo0), (dx1, dg1, db1, dc1, xo1) in zip(ref, outs):
        assert torch.equal(dx0, dx1)
        for a, b in ((dg0, dg1), (db0, db1), (dc0, dc1)):
            assert rel(host(b), host(a)) < 1e-6
        assert torch.equal(xo0, xo1)  # one adder per column, fixed order



@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("M,N,K,cus", [(320 * 7 + 33, 512, 768, 5), (1999, 768, 128, 3), (25216 // 4, 1024, 192, 16), (3300, 256, 1024, 2),
                                       (320 * 40, 2048, 256, 0)])
def test_gemm_persistent_320(ops, epi, M, N, K, cus):
    """Tile 22 = the 320 x 256 ping-pong tile as a persistent grid (gemm_tn_pp320p_kernel): one workgroup per CU walks several tiles and
    the next tile's first K-tile lands under the current tile's last phases and epilogue.  With cu_budget = 2..16 a handful of
    workgroups take 3-20 tiles each (odd and even K-tile counts per tile: the two LDS buffers swap roles from tile to tile; a ragged
    last row tile; the dummy prefetch behind a workgroup's last tile); the last case is a real grid of 256 workgroups.  Same K order and
    epilogue code as tile 21: every output must be BITWISE that of the one-tile kernel, for every epilogue."""
    rng = np.random.default_rng(M + N + K + epi)
    A, Bt = dev(_mk(rng, M, K), bf16), dev(_mk(rng, N, K, 1 / np.sqrt(K)), bf16)
    bias = dev((0.1 * rng.standard_normal(N)).astype(np.float32))
    L = ops._lib.load()

    def run(tile, budget):
        kw = {}
        if epi in (0, 1, 3):
            C = torch.full((M, N), float("nan"), dtype=bf16, device="cuda")
        else:
            C = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        outs = [C]
        if epi == 0:
            kw = dict(alpha=0.125, alpha_cols=N // 2)
        elif epi == 1:
            C2 = torch.full((M, N), float("nan"), dtype=bf16, device="cuda")
            kw = dict(C2=C2, bias=bias)
            outs.append(C2)
        elif epi == 2:
            torch.manual_seed(1)
            kw = dict(bias=bias, aux=torch.randn(M, N, device="cuda"), colscale=bias.abs() + 0.5)
        elif epi == 3:
            torch.manual_seed(2)
            rows = int(L.savit_gemm_colsum_rows(M, N, K, tile))
            slab = torch.full((rows, N), float("nan"), dtype=torch.float32, device="cuda")
            kw = dict(aux=torch.randn(M, N, device="cuda").to(bf16), colsum=slab)
            outs.append(slab)
        else:
            kw = dict(bias=bias, round_out_bf16=True)
        ops.gemm_tn(A, Bt, C, epi, tile=tile, cu_budget=budget, **kw)
        torch.cuda.synchronize()
        return outs

    ref = run(21, 0)
    got = run(22, cus)
    for a, b in zip(ref, got):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a, b)
    if epi == 0:  # and the one-tile kernel against fp64 on this shape
        want = host(A).astype(np.float64) @ host(Bt).astype(np.float64).T
        want[:, :N // 2] *= 0.125
        assert rel(host(ref[0]), rb(want)) < 1e-3
