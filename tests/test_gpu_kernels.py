"""GPU (-m gpu): each gfx950 kernel, through the C ABI debug taps (include/wt_debug.h),
against a float64 numpy reference of the same op.  fp32 MFMA is a k-ordered fmaf chain, so
the error budget is a few 1e-7 * sum|a*b|; tolerances are stated per test."""
import numpy as np
import pytest
from scipy.special import erf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(pkg, assets):
    prefix, vocab = assets("micro")
    e = pkg.Engine(prefix, vocab, True)
    yield e
    e.close()


def gelu(x):
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def rel_err(a, ref):
    return np.abs(a - ref).max() / max(1e-30, np.abs(ref).max())


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 256, 96), (1, 128, 64), (257, 384, 416), (1500, 384, 1152)])
def test_gemm_plain(eng, M, N, K):
    rng = np.random.default_rng(M * 7 + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((N, K)).astype(np.float32)
    C = eng.dbg_gemm(A, W)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    assert rel_err(C, ref) < 2e-6


GEMM_VARIANTS = {0: "fp32 MFMA 128x128", 13: "bf16x3 split k16", 16: "bf16x3 split k16, 2 blocks/CU"}


@pytest.fixture
def gemm_variant(eng):
    def use(v):
        eng.set_option("gemm_variant", v)
    yield use
    eng.set_option("gemm_variant", -1)


@pytest.mark.parametrize("variant", sorted(GEMM_VARIANTS))
def test_gemm_variants_have_fp32_error(eng, gemm_variant, variant):
    """The fp32-storage GEMMs (the fp32-MFMA form and the full-range form that runs the product as six bf16 plane
    products: front end, per-contraction fall-back) must stay inside the SAME fp32 error budget against fp64;
    also on operands with 12 decades of dynamic range and on small-magnitude operands."""
    gemm_variant(variant)
    rng = np.random.default_rng(variant)
    for M, N, K in [(257, 384, 416), (128, 128, 1536), (1500, 384, 1152)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64).T
        assert rel_err(eng.dbg_gemm(A, W), ref) < 2e-6, (variant, M, N, K)
    for sa, sw in ((1e-3, 1.0), (1.0, 1e-4), (300.0, 20.0)):  # magnitudes far from 1 (fp16 alone would lose them)
        A = (rng.standard_normal((128, 384)) * sa).astype(np.float32)
        W = (rng.standard_normal((128, 384)) * sw).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64).T
        assert rel_err(eng.dbg_gemm(A, W), ref) < 2e-6, (variant, sa, sw)
    A = (rng.standard_normal((256, 256)) * 10.0 ** rng.uniform(-6, 6, (256, 256))).astype(np.float32)
    W = (rng.standard_normal((128, 256)) * 10.0 ** rng.uniform(-6, 6, (128, 256))).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    bound = np.abs(A.astype(np.float64)) @ np.abs(W.astype(np.float64)).T  # sum |a||b|: the fp32 error scale
    assert (np.abs(eng.dbg_gemm(A, W) - ref) / bound).max() < 1.5e-6, variant  # fp32 accumulation over K = 256


@pytest.mark.parametrize("M,N,K", [(192, 128, 32), (300, 256, 96), (1, 128, 64), (257, 384, 416), (1500, 384, 1152), (3000, 1152, 384)])
def test_plane_gemm_plain_and_plane_output(eng, M, N, K):
    """The default encoder GEMM: operands as two fp16 planes (split on the host here, by the producing kernels in the
    engine), three f16 MFMA products, fp32 accumulation.  Same fp32 error budget as the fp32-MFMA kernel; the plane
    OUTPUT (what the next contraction reads) reconstructs the result to 22 bits."""
    rng = np.random.default_rng(M * 7 + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + bias
    assert rel_err(eng.dbg_gemm_planes(A, W, bias, epi=1), ref) < 2e-6
    assert rel_err(eng.dbg_gemm_planes(A, W, bias, epi=3, planes_out=True), gelu(ref)) < 4e-6  # output scale is a loose bound here


@pytest.mark.parametrize("epi", [1, 5, 11])
def test_plane_gemm_epilogues(eng, epi):
    rng = np.random.default_rng(epi)
    M, N, K, P = 400, 256, 64, 100
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / 8).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    pos = rng.standard_normal((P, N)).astype(np.float32)
    C = eng.dbg_gemm_planes(A, W, bias=bias, R=R if epi & 4 else None, pos=pos if epi & 8 else None, epi=epi)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + bias
    if epi & 2:
        ref = gelu(ref)
    if epi & 8:
        ref = ref + pos[np.arange(M) % P]
    if epi & 4:
        ref = ref + R
    assert rel_err(C, ref) < 3e-6


@pytest.mark.parametrize("M,K,epi,n_cu", [(500, 384, 5, 0), (700, 1536, 5, 224), (3000, 96, 11, 0), (193, 384, 5, 0),
                                          (48000, 384, 5, 224), (48000, 384, 5, 256)])
def test_plane_gemm_with_fused_layernorm(eng, M, K, epi, n_cu):
    """The N = 384 residual GEMMs (out-projection, fc2) and conv2 write the NEXT LayerNorm's planes from their epilogue
    (the 384-column tile owns whole rows: two-pass statistics exchanged between the four wavefront columns through LDS).
    Checked against fp64: C as before, LayerNorm(C) * g + b to the LayerNorm kernel's own tolerance, the fp32 copy, and
    both 384-column tile heights (192 rows; 256 rows, which the 224-CU pipelined stream picks at 48000 rows)."""
    rng = np.random.default_rng(M + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((384, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(384).astype(np.float32)
    R = (rng.standard_normal((M, 384)) * 3 + 0.7).astype(np.float32) if epi & 4 else None
    pos = rng.standard_normal((150, 384)).astype(np.float32) if epi & 8 else None
    g = (1.0 + 0.3 * rng.standard_normal(384)).astype(np.float32)
    b = rng.standard_normal(384).astype(np.float32)
    if M > 10000:  # tile heights at the engine's size: force the 384-column tile the cost model would pick there
        A[:] = A[:1000].repeat(48, axis=0)[:M]
    C, ln, y32, fused = eng.dbg_gemm_planes_ln(A, W, bias, g, b, R=R, pos=pos, epi=epi, n_cu=n_cu)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + bias
    if epi & 2:
        ref = gelu(ref)
    if epi & 8:
        ref = ref + pos[np.arange(M) % 150]
    if epi & 4:
        ref = ref + R
    assert rel_err(C, ref) < 3e-6
    if M > 10000:
        assert fused  # at the engine's size a 384-column tile is always the choice
    if not fused:  # small M: the cost model may prefer the narrow tile, and the engine then launches the LayerNorm
        return     # kernel itself (covered by the path tests)
    lnref = (ref - ref.mean(1, keepdims=True)) / np.sqrt(ref.var(1, keepdims=True) + 1e-5) * g + b
    assert np.abs(ln - lnref).max() < 1e-5 and np.abs(y32 - lnref).max() < 1e-5
    assert np.abs(ln - y32).max() < 4e-6  # the planes carry 22 bits of the same values


def test_plane_gemm_is_exact_on_integers(eng):
    """A = I (padded) against an ASYMMETRIC integer W catches a swapped C/D row-col map or a wrong LDS swizzle."""
    K, N = 128, 256
    W = ((np.arange(N)[:, None] * 3 + np.arange(K)[None, :] * 7) % 251).astype(np.float32)
    A = np.zeros((K + 70, K), np.float32)
    A[np.arange(K), np.arange(K)] = 1.0
    A[K:, :] = (np.arange(70)[:, None] % 5 - 2 + (np.arange(K)[None, :] % 3)).astype(np.float32)
    C = eng.dbg_gemm_planes(A, W, epi=1)
    assert np.array_equal(C, (A.astype(np.float64) @ W.astype(np.float64).T).astype(np.float32))


def test_gemm_split_and_fp32_mfma_agree_to_rounding(eng, gemm_variant):
    """The split kernel and the fp32-MFMA kernel differ only by accumulation-order rounding."""
    rng = np.random.default_rng(99)
    A = rng.standard_normal((512, 384)).astype(np.float32)
    W = (rng.standard_normal((384, 384)) / 20).astype(np.float32)
    gemm_variant(0)
    C0 = eng.dbg_gemm(A, W)
    for v in (13, 16):
        gemm_variant(v)
        C1 = eng.dbg_gemm(A, W)
        assert np.abs(C0 - C1).max() < 4e-6 * np.abs(C0).max(), v


def test_gemm_is_exact_on_integers(eng):
    """A = I (padded) against an ASYMMETRIC integer W catches a swapped C/D row-col map."""
    K, N = 128, 256
    W = (np.arange(N)[:, None] * 3 + np.arange(K)[None, :] * 7).astype(np.float32) % 251
    A = np.zeros((K, K), np.float32)
    A[np.arange(K), np.arange(K)] = 1.0
    C = eng.dbg_gemm(A, W)
    assert np.array_equal(C, W.T)
    A2 = (np.arange(200)[:, None] % 5 - 2 + (np.arange(K)[None, :] % 3)).astype(np.float32)
    assert np.array_equal(eng.dbg_gemm(A2, W), (A2.astype(np.float64) @ W.astype(np.float64).T).astype(np.float32))


@pytest.mark.parametrize("epi", [1, 3, 5, 11])
def test_gemm_epilogues(eng, epi):
    rng = np.random.default_rng(epi)
    M, N, K, P = 400, 256, 64, 100
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / 8).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    pos = rng.standard_normal((P, N)).astype(np.float32)
    C = eng.dbg_gemm(A, W, bias=bias, R=R if epi & 4 else None, pos=pos if epi & 8 else None, epi=epi)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + bias
    if epi & 2:
        ref = gelu(ref)
    if epi & 8:
        ref = ref + pos[np.arange(M) % P]
    if epi & 4:
        ref = ref + R
    assert rel_err(C, ref) < 3e-6


@pytest.mark.parametrize("B,N,K", [(32, 384, 384), (32, 128, 512), (7, 1536, 384), (64, 384, 1536), (33, 1000, 128), (32, 51865, 384),
                                   (128, 1152, 384), (100, 512, 512), (128, 96, 128)])
def test_decoder_gemm_and_argmax(eng, B, N, K):
    rng = np.random.default_rng(B + N + K)
    X = rng.standard_normal((B, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / 16).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = X.astype(np.float64) @ W.astype(np.float64).T
    assert rel_err(eng.dbg_dec_gemm(X, W, bias, mode=0), ref + bias) < 3e-6
    assert rel_err(eng.dbg_dec_gemm(X, W, bias, mode=1), gelu(ref + bias)) < 3e-6
    Y, am = eng.dbg_dec_gemm(X, W, mode=3)
    assert rel_err(Y, ref) < 3e-6
    # the fused argmax is exact w.r.t. the values the kernel itself produced, last index on ties
    assert list(am) == [int(N - 1 - np.argmax(Y[b][::-1])) for b in range(B)]


@pytest.mark.parametrize("B,N,K", [(32, 384, 384), (32, 384, 1536), (5, 128, 128), (64, 512, 2048), (128, 384, 384), (128, 384, 1536)])
def test_decoder_gemm_residual_in_place(eng, B, N, K):
    rng = np.random.default_rng(B * N + K)
    X = rng.standard_normal((B, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / 16).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((B, N)).astype(np.float32)
    Y = eng.dbg_dec_gemm(X, W, bias, mode=2, R=R)
    assert rel_err(Y, R + bias + X.astype(np.float64) @ W.astype(np.float64).T) < 3e-6
    assert np.array_equal(Y, eng.dbg_dec_gemm(X, W, bias, mode=2, R=R))  # fixed reduction order, no atomics


def test_decoder_gemm_dynamic_scale_covers_any_magnitude(eng):
    """The activation planes take a power-of-two scale per (row, k-slice) from the data: rows of magnitude 1e-6 next
    to rows of 1e+6, a row that is zero in one k-slice, and 12 decades inside a row stay at fp32-level error
    (relative to sum |x||w|, the fp32 error scale)."""
    rng = np.random.default_rng(17)
    B, N, K = 32, 384, 384
    X = rng.standard_normal((B, K)).astype(np.float32)
    X[0] *= 1e-6
    X[1] *= 1e6
    X[2, :96] = 0.0
    X[3] = (rng.standard_normal(K) * 10.0 ** rng.uniform(-6, 6, K)).astype(np.float32)
    X[4] = 0.0
    W = (rng.standard_normal((N, K)) / 16 * 10.0 ** rng.uniform(-2, 2, (N, 1))).astype(np.float32)
    Y = eng.dbg_dec_gemm(X, W, np.zeros(N, np.float32), mode=0)
    ref = X.astype(np.float64) @ W.astype(np.float64).T
    bound = np.abs(X.astype(np.float64)) @ np.abs(W.astype(np.float64)).T + 1e-300
    assert (np.abs(Y - ref) / bound).max() < 1.5e-6
    assert np.all(Y[4] == 0.0)


def test_decoder_argmax_tie_rule(eng):
    """Duplicate rows of W give bit-identical logits: the LAST index must win (whisper.cpp:353)."""
    rng = np.random.default_rng(5)
    B, N, K = 4, 512, 128
    X = rng.standard_normal((B, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / 64).astype(np.float32)
    W[100] = W[7] = W[400] = X[0] / 4  # large identical logit at 7, 100, 400 for row 0
    W[33] = W[300] = X[1] / 4
    Y, am = eng.dbg_dec_gemm(X, W, mode=3)
    assert Y[0, 7] == Y[0, 100] == Y[0, 400] and am[0] == 400
    assert am[1] == 300


@pytest.mark.parametrize("B,K,N,gelu_on", [(32, 384, 1152, False), (9, 128, 512, True), (64, 512, 512, False)])
def test_decoder_ln_fused_gemm(eng, B, K, N, gelu_on):
    rng = np.random.default_rng(B + K + N)
    xin = (rng.standard_normal((B, K)) * 2 + 0.5).astype(np.float32)
    g_, b_ = rng.standard_normal(K).astype(np.float32), rng.standard_normal(K).astype(np.float32)
    W = (rng.standard_normal((N, K)) / 16).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    Y, _ = eng.dbg_dec_ln_gemm(W, bias, g_, b_, xin=xin, gelu=gelu_on)
    x = xin.astype(np.float64)
    ln = (x - x.mean(1, keepdims=True)) / np.sqrt(x.var(1, keepdims=True) + 1e-5) * g_ + b_
    ref = ln @ W.astype(np.float64).T + bias
    assert rel_err(Y, gelu(ref) if gelu_on else ref) < 5e-6


def test_decoder_ln_fused_embedding(eng):
    rng = np.random.default_rng(8)
    B, K, N, V = 6, 128, 384, 50
    tok = rng.standard_normal((V, K)).astype(np.float32)
    pos = rng.standard_normal((10, K)).astype(np.float32)
    ids = np.array([3, 49, 0, 7, 7, 12], np.int64)
    g_, b_ = np.ones(K, np.float32), np.zeros(K, np.float32)
    W = (rng.standard_normal((N, K)) / 16).astype(np.float32)
    bias = np.zeros(N, np.float32)
    Y, xout = eng.dbg_dec_ln_gemm(W, bias, g_, b_, ids=ids, pos=4, tok_emb=tok, pos_emb=pos)
    assert np.array_equal(xout, tok[ids] + pos[4])


@pytest.mark.parametrize("M,d", [(5, 128), (1000, 384), (33, 512)])
def test_layernorm(eng, M, d):
    rng = np.random.default_rng(M + d)
    x = (rng.standard_normal((M, d)) * 3 + 1).astype(np.float32)
    g = rng.standard_normal(d).astype(np.float32)
    b = rng.standard_normal(d).astype(np.float32)
    y = eng.dbg_layernorm(x, g, b)
    xd = x.astype(np.float64)
    ref = (xd - xd.mean(1, keepdims=True)) / np.sqrt(xd.var(1, keepdims=True) + 1e-5) * g + b
    assert np.abs(y - ref).max() < 5e-6


def attn_ref(q, k, v, mask_from=None):
    s = (q @ k.T) / 8.0
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    return (p / p.sum(-1, keepdims=True)) @ v


@pytest.mark.parametrize("attn_variant", [0, 1])  # fp32-storage forms: 0 = fp32 MFMA; 1 = three bf16 planes (the fall-back)
@pytest.mark.parametrize("B,T,H", [(1, 64, 1), (2, 100, 2), (1, 1500, 6), (3, 333, 2)])
def test_encoder_attention(eng, B, T, H, attn_variant):
    eng.set_option("attn_variant", attn_variant)
    rng = np.random.default_rng(B * 1000 + T + H)
    d = 64 * H
    qkv = rng.standard_normal((B * T, 3 * d)).astype(np.float32)
    out = eng.dbg_encoder_attention(qkv, B, T, H)
    eng.set_option("attn_variant", 4)
    q64 = qkv.astype(np.float64).reshape(B, T, 3, H, 64)
    for b in range(B):
        for h in range(H):
            ref = attn_ref(q64[b, :, 0, h], q64[b, :, 1, h], q64[b, :, 2, h])
            got = out.reshape(B, T, H, 64)[b, :, h]
            assert np.abs(got - ref).max() < 2e-5, (b, h)


@pytest.mark.parametrize("B,T,H", [(1, 64, 1), (2, 100, 2), (1, 1500, 6), (3, 333, 2), (1, 128, 1)])
def test_encoder_attention_planes(eng, B, T, H):
    """The default encoder attention: q, k, v as fp16 planes, V^T fragments by ds_read_b64_tr_b16."""
    rng = np.random.default_rng(B * 1000 + T + H)
    d = 64 * H
    qkv = rng.standard_normal((B * T, 3 * d)).astype(np.float32)
    out = eng.dbg_encoder_attention_planes(qkv, B, T, H)
    q = qkv.astype(np.float64).reshape(B, T, 3 * d)
    for b in range(B):
        for h in range(H):
            ref = attn_ref(q[b, :, h * 64:(h + 1) * 64], q[b, :, d + h * 64:d + (h + 1) * 64], q[b, :, 2 * d + h * 64:2 * d + (h + 1) * 64])
            assert np.abs(out.reshape(B, T, d)[b, :, h * 64:(h + 1) * 64] - ref).max() < 2e-5, (b, h)


def test_encoder_attention_planes_forces_rescale(eng):
    """A key far above the others late in the sequence forces the running-max rescale branch."""
    rng = np.random.default_rng(4)
    T, d = 300, 64
    qkv = rng.standard_normal((T, 3 * d)).astype(np.float32)
    qkv[250, d:2 * d] = qkv[7, 0:d] * 6.0  # key 250 aligned with query 7: its score dwarfs the earlier ones
    out = eng.dbg_encoder_attention_planes(qkv, 1, T, 1)
    q = qkv.astype(np.float64)
    ref = attn_ref(q[:, 0:64], q[:, 64:128], q[:, 128:192])
    assert np.abs(out - ref).max() < 2e-5


@pytest.mark.parametrize("attn_variant", [0, 1])
def test_encoder_attention_forces_rescale(eng, attn_variant):
    """Online softmax: spike one key late in the sequence so the running max jumps at a chosen
    tile (the rare branch), and check the FULL tensor against fp64."""
    rng = np.random.default_rng(77)
    B, T, H = 1, 300, 1
    qkv = rng.standard_normal((T, 192)).astype(np.float32)
    qkv[250, 64:128] = qkv[10, 0:64] * 6.0  # key 250 lines up with query 10: score >> others
    qkv[120, 64:128] = qkv[11, 0:64] * 4.0
    eng.set_option("attn_variant", attn_variant)
    out = eng.dbg_encoder_attention(qkv, B, T, H)
    eng.set_option("attn_variant", 4)
    q = qkv.astype(np.float64)
    ref = attn_ref(q[:, 0:64], q[:, 64:128], q[:, 128:192])
    assert np.abs(out - ref).max() < 2e-5


def _ln64(x, g, b):
    x = x.astype(np.float64)
    return (x - x.mean(1, keepdims=True)) / np.sqrt(x.var(1, keepdims=True) + 1e-5) * g + b


@pytest.mark.parametrize("B,H,T,chunks,nq", [(2, 2, 100, 4, 1), (3, 6, 1500, 2, 1), (1, 2, 1500, 1, 1), (2, 2, 37, 8, 1),
                                             (32, 6, 1500, 2, 4), (5, 8, 200, 2, 3), (3, 2, 64, 4, 2)])
def test_cross_attention_with_fused_query_projection(eng, B, H, T, chunks, nq):
    """cross_attention_step makes its own queries: q = LayerNorm(x[row]) . Wq^T + bq per (clip, head), nq rows per clip
    (row = p * B + b: the prompt positions of the first decoder pass share one sweep of the cache)."""
    rng = np.random.default_rng(B + H + T + chunks + nq)
    d = H * 64
    x = (rng.standard_normal((nq * B, d)) * 2 + 0.3).astype(np.float32)
    g_, b_ = (1 + 0.2 * rng.standard_normal(d)).astype(np.float32), (0.1 * rng.standard_normal(d)).astype(np.float32)
    wq = (rng.standard_normal((d, d)) / np.sqrt(d)).astype(np.float32)
    bq = (0.1 * rng.standard_normal(d)).astype(np.float32)
    kc = rng.standard_normal((B, H, T, 64)).astype(np.float32)
    vc = rng.standard_normal((B, H, T, 64)).astype(np.float32)
    out = eng.dbg_cross_attention(x, g_, b_, wq, bq, kc, vc, chunks, nq)
    q = _ln64(x, g_, b_) @ wq.astype(np.float64).T + bq
    for p in range(nq):
        for b in range(B if B <= 5 else 3):
            for h in range(H):
                sl = slice(h * 64, (h + 1) * 64)
                ref = attn_ref(q[p * B + b, sl][None], kc[b, h].astype(np.float64), vc[b, h].astype(np.float64))[0]
                assert np.abs(out[p * B + b, sl] - ref).max() < 2e-5, (p, b, h)


def test_self_attention_appends_and_attends(eng):
    rng = np.random.default_rng(3)
    B, H, cap = 3, 2, 32
    d = 64 * H
    kc = np.zeros((B, cap, d), np.float32)
    vc = np.zeros((B, cap, d), np.float32)
    for pos in range(6):
        qkv = rng.standard_normal((B, 3 * d)).astype(np.float32)
        out, kc2, vc2 = eng.dbg_self_attention(qkv, kc, vc, pos)
        kc[:, pos] = qkv[:, d:2 * d]
        vc[:, pos] = qkv[:, 2 * d:]
        assert np.array_equal(kc2, kc) and np.array_equal(vc2, vc)
        for b in range(B):
            for h in range(H):
                sl = slice(h * 64, (h + 1) * 64)
                ref = attn_ref(qkv[b, sl].astype(np.float64)[None], kc[b, :pos + 1, sl].astype(np.float64),
                               vc[b, :pos + 1, sl].astype(np.float64))[0]
                assert np.abs(out[b, sl] - ref).max() < 1e-5


@pytest.mark.parametrize("pos0,npos", [(0, 4), (0, 2), (3, 3), (28, 4)])
def test_self_attention_several_positions_in_one_pass(eng, pos0, npos):
    """The prompt pass: npos new positions per clip (rows p * B + b), causal among themselves and over the cache."""
    rng = np.random.default_rng(pos0 * 10 + npos)
    B, H, cap = 5, 2, 32
    d = 64 * H
    kc = np.zeros((B, cap, d), np.float32)
    vc = np.zeros((B, cap, d), np.float32)
    kc[:, :pos0] = rng.standard_normal((B, pos0, d))
    vc[:, :pos0] = rng.standard_normal((B, pos0, d))
    qkv = rng.standard_normal((npos * B, 3 * d)).astype(np.float32)
    out, kc2, vc2 = eng.dbg_self_attention(qkv, kc, vc, pos0, npos)
    for p in range(npos):
        kc[:, pos0 + p] = qkv[p * B:(p + 1) * B, d:2 * d]
        vc[:, pos0 + p] = qkv[p * B:(p + 1) * B, 2 * d:]
    assert np.array_equal(kc2, kc) and np.array_equal(vc2, vc)
    for p in range(npos):
        for b in range(B):
            for h in range(H):
                sl = slice(h * 64, (h + 1) * 64)
                n = pos0 + p + 1
                ref = attn_ref(qkv[p * B + b, sl].astype(np.float64)[None], kc[b, :n, sl].astype(np.float64),
                               vc[b, :n, sl].astype(np.float64))[0]
                assert np.abs(out[p * B + b, sl] - ref).max() < 1e-5, (p, b, h)


# ---------------------------------------------------------------- bf16 storage mode (BASELINE configs[3]) ---

def bf16_round(x):
    """round-to-nearest-even to bf16, returned as float32 (what the bf16 storage mode keeps in HBM)"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("M,N,K", [(192, 128, 64), (300, 256, 128), (1, 128, 64), (257, 384, 448), (1500, 512, 1536), (3000, 1536, 512),
                                   (129, 256, 64), (193, 128, 64), (500, 640, 192), (1000, 2048, 512), (77, 768, 64)])
def test_bf16_gemm_matches_bf16_rounded_operands(eng, M, N, K):
    """gemm_bf16_planes contracts bf16 operands EXACTLY (bf16 x bf16 products are exact in fp32) with fp32
    accumulation: against fp64 on the rounded operands the error is the fp32 accumulation error, not a bf16 one.
    The bf16 OUTPUT is that result rounded once more (half an ulp of bf16 = 2^-9 relative)."""
    rng = np.random.default_rng(M + 3 * N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = bf16_round(A).astype(np.float64) @ bf16_round(W).astype(np.float64).T + bias
    assert rel_err(eng.dbg_gemm_bf16(A, W, bias, epi=1), ref) < 2e-6
    out16 = eng.dbg_gemm_bf16(A, W, bias, epi=3, bf16_out=True)
    g = gelu(ref)
    assert np.array_equal(out16, bf16_round(out16))  # values on the bf16 grid
    assert (np.abs(out16 - g) <= np.abs(g) * 2.0 ** -8 + 1e-6).all()


@pytest.mark.parametrize("epi", [1, 5, 11])
def test_bf16_gemm_epilogues(eng, epi):
    rng = np.random.default_rng(100 + epi)
    M, N, K, P = 400, 256, 64, 100
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / 8).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    pos = rng.standard_normal((P, N)).astype(np.float32)
    C = eng.dbg_gemm_bf16(A, W, bias=bias, R=R if epi & 4 else None, pos=pos if epi & 8 else None, epi=epi)
    ref = bf16_round(A).astype(np.float64) @ bf16_round(W).astype(np.float64).T + bias
    if epi & 2:
        ref = gelu(ref)
    if epi & 8:
        ref = ref + pos[np.arange(M) % P]
    if epi & 4:
        ref = ref + R
    assert rel_err(C, ref) < 3e-6


@pytest.mark.parametrize("M,N,K", [(192, 128, 64), (500, 128, 512), (256, 384, 384), (1000, 384, 1536), (128, 512, 512), (777, 512, 2048), (1, 512, 64),
                                   (300, 256, 128)])
def test_bf16_gemm_layernorm_fused_in_the_epilogue(eng, M, N, K):
    """Round 4: the bf16 mode's out-projection / fc2 / conv2 GEMMs write LayerNorm(finished row) as the next GEMM's bf16
    operand (and, for ln_post, as fp32) from their epilogue — whole-row tiles for d_model 128 / 384 / 512, rows past M in
    the last tile, one to four wavefront columns exchanging the row statistics.  Against fp64 on the bf16-rounded operands:
    x at the GEMM's bar, the fp32 LayerNorm at 2e-5 absolute, the plane = that rounded to bf16 (half an ulp on top).  N = 256
    has no whole-row tile: the launcher says so and the caller keeps the separate LayerNorm launch."""
    rng = np.random.default_rng(M + 7 * N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = (rng.standard_normal((M, N)) * 2).astype(np.float32)
    g = (1.0 + 0.3 * rng.standard_normal(N)).astype(np.float32)
    b = (0.2 * rng.standard_normal(N)).astype(np.float32)
    x, ln, y32, fused = eng.dbg_gemm_bf16_ln(A, W, bias, R, g, b)
    ref = R.astype(np.float64) + bias + bf16_round(A).astype(np.float64) @ bf16_round(W).astype(np.float64).T
    assert rel_err(x, ref) < 3e-6
    assert fused == (N in (128, 384, 512))
    if not fused:
        return
    mu = ref.mean(axis=1, keepdims=True)
    var = ((ref - mu) ** 2).mean(axis=1, keepdims=True)
    want = (ref - mu) / np.sqrt(var + 1e-5) * g + b
    assert np.abs(y32 - want).max() < 2e-5
    assert np.array_equal(ln, bf16_round(ln))
    assert (np.abs(ln - want) <= np.abs(want) * 2.0 ** -8 + 3e-5).all()


def test_bf16_gemm_is_exact_on_integers(eng):
    """Small integers are exact in bf16: an asymmetric integer W against an identity-plus-pattern A catches a swapped
    fragment map or a wrong LDS swizzle of the 128-byte-row image."""
    K, N = 128, 256
    W = ((np.arange(N)[:, None] * 3 + np.arange(K)[None, :] * 7) % 127).astype(np.float32)
    A = np.zeros((K + 70, K), np.float32)
    A[np.arange(K), np.arange(K)] = 1.0
    A[K:, :] = (np.arange(70)[:, None] % 5 - 2 + (np.arange(K)[None, :] % 3)).astype(np.float32)
    C = eng.dbg_gemm_bf16(A, W, epi=1)
    assert np.array_equal(C, (A.astype(np.float64) @ W.astype(np.float64).T).astype(np.float32))
    # the other two tile shapes (192 x 384 and 192 x 128; 192 x 256 above)
    for N2 in (384, 128):
        W2 = ((np.arange(N2)[:, None] * 5 + np.arange(K)[None, :] * 11) % 113).astype(np.float32)
        C2 = eng.dbg_gemm_bf16(A, W2, epi=1)
        assert np.array_equal(C2, (A.astype(np.float64) @ W2.astype(np.float64).T).astype(np.float32))


@pytest.mark.parametrize("growth", [0.5, 2.9, 3.1, 9.0])
def test_encoder_attention_planes_deferred_maximum(eng, growth):
    """The kernel raises its running maximum only when a tile's maximum exceeds it by more than 3 (log2 domain); below
    that the tile's probabilities exceed 1 (up to 2^3) and must still come out right — also when the excess accumulates
    over several tiles before the rescale fires.  Scores that grow by `growth` (log2 units) per 64-key tile, against
    the full fp64 softmax: just under and just over the threshold, far under, far over."""
    rng = np.random.default_rng(int(growth * 10))
    T, d = 400, 64
    qkv = (rng.standard_normal((T, 3 * d)) * 0.05).astype(np.float32)
    qkv[:, 0:d] += 1.0                                           # q ~ ones: |q|^2 ~ 64
    ramp = growth / (64 * 1.4426950408889634) * np.arange(T)     # score of key j ~ ramp[j] after the 1/8 scaling
    qkv[:, d:2 * d] += (ramp / 8.0)[:, None].astype(np.float32)  # k_j ~ ramp_j / 8 * ones -> q.k / 8 = ramp_j
    out = eng.dbg_encoder_attention_planes(qkv, 1, T, 1)
    q = qkv.astype(np.float64)
    ref = attn_ref(q[:, 0:64], q[:, 64:128], q[:, 128:192])
    assert np.abs(out - ref).max() < 2e-5, growth


@pytest.mark.parametrize("B,T,H", [(1, 64, 1), (2, 100, 2), (1, 1500, 8), (3, 333, 2)])
def test_encoder_attention_bf16_storage(eng, B, T, H):
    """encoder_attention_planes<true>: q, k, v read as bf16, probabilities rounded to bf16 for the PV product, fp32
    accumulation and softmax statistics.  Against fp64 attention on the bf16-rounded inputs what is left is the
    rounding of the probabilities (2^-9 relative each, averaging down over the keys) and of the bf16 output."""
    rng = np.random.default_rng(B * 1000 + T + H)
    d = 64 * H
    qkv = rng.standard_normal((B * T, 3 * d)).astype(np.float32)
    out = eng.dbg_encoder_attention_bf16(qkv, B, T, H)
    q = bf16_round(qkv).astype(np.float64).reshape(B, T, 3 * d)
    for b in range(B):
        for h in range(H):
            ref = attn_ref(q[b, :, h * 64:(h + 1) * 64], q[b, :, d + h * 64:d + (h + 1) * 64], q[b, :, 2 * d + h * 64:2 * d + (h + 1) * 64])
            err = np.abs(out.reshape(B, T, d)[b, :, h * 64:(h + 1) * 64] - ref)
            assert err.max() < 2.0 ** -8 * np.abs(ref).max() + 4e-3, (b, h, err.max())


@pytest.mark.parametrize("n_cu,expect", [(8, "256-row"), (11, "192-row"), (3, "256-row, three rounds")])
def test_plane_gemm_every_tile_shape_gives_the_same_result(pkg, eng, n_cu, expect):
    """The plane GEMM picks its tile (192 x 128, 192 x 384, 256 x 384) from how the blocks fill the CUs the stream may
    use.  With M = 2000, N = 384: 11 row tiles of 192 or 8 of 256 — on 8 CUs the 256-row tile wins (one round instead of
    two), on 11 CUs the 192-row one.  Whatever is picked, the result is the fp32-accurate product.  The tiles agree
    with each other to accumulation rounding: the 192 x 384 tile runs on 16 x 16 x 32 MFMAs since round 4 (a 32-deep
    sum per instruction), the others on 32 x 32 x 16 (16-deep) — same k order, different partial sums, so the last
    bits may differ; every schedule of ONE MFMA shape is bit-identical (wt_dbg_set_plane_gemm_mode 0 and 1)."""
    rng = np.random.default_rng(2000)
    M, N, K = 2000, 384, 384
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + bias
    base = eng.dbg_gemm_planes(A, W, bias, epi=1)
    got = eng.dbg_gemm_planes(A, W, bias, epi=1, n_cu=n_cu)
    assert rel_err(got, ref) < 2e-6, expect
    assert rel_err(got, base) < 1e-6 and np.abs(got - base).max() < 4e-6, expect
    L = pkg.lib()
    try:  # the two schedules of the 32 x 32 x 16 form of the 192 x 384 tile: same arithmetic, bit for bit
        outs = []
        for mode in (0, 1):
            assert L.wt_dbg_set_plane_gemm_mode(mode) == 0
            outs.append(eng.dbg_gemm_planes(A, W, bias, epi=1, n_cu=11))
        assert np.array_equal(outs[0], outs[1])
        assert rel_err(outs[0], ref) < 2e-6
    finally:
        assert L.wt_dbg_set_plane_gemm_mode(2) == 0
    # several tiles per CU with plane output: the PERSISTENT form of the ping-pong kernel (a block walks its tiles and
    # prefetches the next one's first k-tile from its epilogue) — bit for bit the one-tile-per-block kernel (mode 4),
    # whole tiles and a ragged last row tile, both epilogues it serves
    for Mp, Np, Kp, epi in ((3000, 1152, 384, 1), (2900, 1536, 384, 3), (5000, 768, 256, 1)):
        Ap = rng.standard_normal((Mp, Kp)).astype(np.float32)
        Wp = (rng.standard_normal((Np, Kp)) / np.sqrt(Kp)).astype(np.float32)
        bp = rng.standard_normal(Np).astype(np.float32)
        try:
            assert L.wt_dbg_set_plane_gemm_mode(4) == 0
            one = eng.dbg_gemm_planes(Ap, Wp, bp, epi=epi, planes_out=True, n_cu=8)
        finally:
            assert L.wt_dbg_set_plane_gemm_mode(2) == 0
        per = eng.dbg_gemm_planes(Ap, Wp, bp, epi=epi, planes_out=True, n_cu=8)
        refp = Ap.astype(np.float64) @ Wp.astype(np.float64).T + bp
        assert np.array_equal(one, per), (Mp, Np, Kp)
        assert rel_err(per, gelu(refp) if epi & 2 else refp) < 4e-6
    assert rel_err(eng.dbg_gemm_planes(A, W, bias, R=R, epi=5, n_cu=n_cu), ref + R) < 3e-6
    assert rel_err(eng.dbg_gemm_planes(A, W, bias, epi=3, planes_out=True, n_cu=n_cu), gelu(ref)) < 4e-6


@pytest.mark.parametrize("B,H,T,chunks,nq", [(1, 6, 1500, 8, 1), (3, 6, 333, 4, 2), (2, 2, 100, 2, 4), (2, 8, 200, 3, 2),
                                              (1, 6, 64, 2, 1), (2, 6, 1500, 16, 4), (5, 6, 97, 1, 1)])
def test_cross_attention_absorbed(eng, B, H, T, chunks, nq):
    """k_cross_absorbed.hip: per (row, head) Wv_h (sum_j softmax_j(q' . e_j) e_j) + bv_h over the encoder output
    itself (scores in the log2 domain), key chunks combined — against fp64.  Covers d_model 384 / 128 / 512 (6 / 2 / 8 heads), ragged
    last tiles and chunks, one to four positions per clip (positions beyond 16 / heads query columns take a second
    launch), a single chunk, and the full 1500-key sweep."""
    rng = np.random.default_rng(B * 1000 + T + H + nq)
    d = 64 * H
    E = rng.standard_normal((B, T, d)).astype(np.float32)
    qp = (rng.standard_normal((nq * B, H * d)) * (3.0 / np.sqrt(d))).astype(np.float32)  # scores O(3) in log2 units
    qp[0, :d] *= 4.0          # one peaky query column: its running maximum jumps
    wv = (rng.standard_normal((d, d)) / np.sqrt(d)).astype(np.float32)
    bv = rng.standard_normal(d).astype(np.float32)
    out = eng.dbg_cross_absorbed(qp, E, wv, bv, B, H, T, chunks, nq)
    E64 = E.astype(np.float64)
    for r in range(nq * B):
        b = r % B                                   # row = p * B + b
        for h in range(H):
            s = E64[b] @ qp[r, h * d:(h + 1) * d].astype(np.float64)
            p = np.exp2(s - s.max())
            c = (p[:, None] * E64[b]).sum(0) / p.sum()
            ref = wv[h * 64:(h + 1) * 64].astype(np.float64) @ c + bv[h * 64:(h + 1) * 64]
            assert np.abs(out[r, h * 64:(h + 1) * 64] - ref).max() < 2e-5, (r, h)


def _bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


@pytest.mark.parametrize("B,H,T,chunks,nq", [(2, 8, 1500, 4, 2), (3, 6, 333, 4, 2), (2, 2, 100, 2, 4), (1, 8, 64, 2, 1),
                                              (5, 6, 97, 1, 1)])
def test_cross_attention_absorbed_bf16_storage(eng, B, H, T, chunks, nq):
    """The bf16 storage variant of the absorbed cross-attention (BASELINE configs[3]): E is one bf16 plane, queries and
    probabilities are rounded to bf16 in the kernel, one bf16 MFMA per product, fp32 accumulation.  Against fp64 on the
    bf16-ROUNDED E and queries the only remaining difference is the rounding of the probabilities (2^-9 relative each,
    averaging out over the keys) and fp32 accumulation: 4e-3 absolute on O(1) outputs; d_model 512 / 384 / 128."""
    rng = np.random.default_rng(B * 77 + T + H + nq)
    d = 64 * H
    E = _bf16_round(rng.standard_normal((B, T, d)).astype(np.float32))
    qp = _bf16_round((rng.standard_normal((nq * B, H * d)) * (3.0 / np.sqrt(d))).astype(np.float32))
    wv = (rng.standard_normal((d, d)) / np.sqrt(d)).astype(np.float32)
    bv = rng.standard_normal(d).astype(np.float32)
    out = eng.dbg_cross_absorbed(qp, E, wv, bv, B, H, T, chunks, nq, bf16=True)
    E64 = E.astype(np.float64)
    worst = 0.0
    for r in range(nq * B):
        b = r % B
        for h in range(H):
            s = E64[b] @ qp[r, h * d:(h + 1) * d].astype(np.float64)
            p = np.exp2(s - s.max())
            c = (p[:, None] * E64[b]).sum(0) / p.sum()
            ref = wv[h * 64:(h + 1) * 64].astype(np.float64) @ c + bv[h * 64:(h + 1) * 64]
            worst = max(worst, np.abs(out[r, h * 64:(h + 1) * 64] - ref).max())
    assert worst < 4e-3, worst


# ------------------------------------------------ the DECODER's kernels in the bf16 storage mode (round 4) ---
# dec_gemm<..., BF = true>, dec_logits_persistent<..., true, ...>, self_attention_step<true>, cross_attention_step<..., true>:
# weights are one bf16 plane in fragment order, activations are rounded to bf16 in registers (round to nearest even),
# products are exact in fp32 and accumulated in fp32 — so against fp64 on the bf16-ROUNDED operands only fp32
# accumulation error remains, the bar of the fp32-accurate kernels.

@pytest.mark.parametrize("B,N,K", [(32, 512, 512), (64, 1536, 512), (7, 2048, 512), (64, 512, 2048), (33, 1000, 128), (32, 51865, 384),
                                   (64, 51865, 512), (128, 1536, 512), (100, 384, 384), (128, 96, 128)])
def test_decoder_gemm_bf16_storage(eng, B, N, K):
    """Every epilogue of dec_gemm<BF = true> and the persistent logits kernel (N = 51865: whisper's vocabulary, base and
    tiny widths; 64 rows = a pair of batches) with the fused argmax and its tie rule."""
    rng = np.random.default_rng(B + N + K + 1)
    X = rng.standard_normal((B, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / 16).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = _bf16_round(X).astype(np.float64) @ _bf16_round(W).astype(np.float64).T
    assert rel_err(eng.dbg_dec_gemm(X, W, bias, mode=0, bf16=True), ref + bias) < 3e-6
    assert rel_err(eng.dbg_dec_gemm(X, W, bias, mode=1, bf16=True), gelu(ref + bias)) < 3e-6
    if B <= 64 or N * K <= 1536 * 2048:
        R = rng.standard_normal((B, N)).astype(np.float32)
        Y = eng.dbg_dec_gemm(X, W, bias, mode=2, R=R, bf16=True)
        assert rel_err(Y, R + bias + ref) < 3e-6
        assert np.array_equal(Y, eng.dbg_dec_gemm(X, W, bias, mode=2, R=R, bf16=True))  # fixed reduction order
    Y, am = eng.dbg_dec_gemm(X, W, mode=3, bf16=True)
    assert rel_err(Y, ref) < 3e-6
    assert list(am) == [int(N - 1 - np.argmax(Y[b][::-1])) for b in range(B)]
    # it IS the bf16 instantiation: the fp32-accurate kernel on the same operands differs by bf16 rounding
    assert rel_err(eng.dbg_dec_gemm(X, W, bias, mode=0), ref + bias) > 1e-4


def test_decoder_argmax_tie_rule_bf16_storage(eng):
    rng = np.random.default_rng(6)
    B, N, K = 4, 512, 128
    X = _bf16_round(rng.standard_normal((B, K)).astype(np.float32))
    W = (rng.standard_normal((N, K)) / 64).astype(np.float32)
    W[100] = W[7] = W[400] = X[0] / 4
    W[33] = W[300] = X[1] / 4
    Y, am = eng.dbg_dec_gemm(X, W, mode=3, bf16=True)
    assert Y[0, 7] == Y[0, 100] == Y[0, 400] and am[0] == 400
    assert am[1] == 300


@pytest.mark.parametrize("B,K,N,gelu_on", [(32, 512, 1536, False), (9, 128, 512, True), (64, 512, 2048, True), (64, 384, 1152, False)])
def test_decoder_ln_fused_gemm_bf16_storage(eng, B, K, N, gelu_on):
    """LayerNorm prologue in fp32, its rows rounded to bf16 for the matrix cores.  The reference rounds an fp64
    LayerNorm: an element within 1e-7 of a bf16 rounding boundary may round the other way (2^-8 of one product) —
    hence an absolute bar of 1e-3 on O(1) outputs beside the fp32-accumulation bar on the median error."""
    rng = np.random.default_rng(B + K + N + 2)
    xin = (rng.standard_normal((B, K)) * 2 + 0.5).astype(np.float32)
    g_, b_ = rng.standard_normal(K).astype(np.float32), rng.standard_normal(K).astype(np.float32)
    W = (rng.standard_normal((N, K)) / 16).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    Y, _ = eng.dbg_dec_ln_gemm(W, bias, g_, b_, xin=xin, gelu=gelu_on, bf16=True)
    x = xin.astype(np.float64)
    ln = (x - x.mean(1, keepdims=True)) / np.sqrt(x.var(1, keepdims=True) + 1e-5) * g_ + b_
    ref = _bf16_round(ln.astype(np.float32)).astype(np.float64) @ _bf16_round(W).astype(np.float64).T + bias
    if gelu_on:
        ref = gelu(ref)
    err = np.abs(Y - ref)
    assert err.max() < 1e-3 and np.median(err) < 2e-6, (err.max(), np.median(err))


def test_decoder_ln_fused_embedding_bf16_storage(eng):
    rng = np.random.default_rng(9)
    B, K, N, V = 6, 128, 384, 50
    tok = rng.standard_normal((V, K)).astype(np.float32)
    pos = rng.standard_normal((10, K)).astype(np.float32)
    ids = np.array([3, 49, 0, 7, 7, 12], np.int64)
    g_, b_ = np.ones(K, np.float32), np.zeros(K, np.float32)
    W = (rng.standard_normal((N, K)) / 16).astype(np.float32)
    Y, xout = eng.dbg_dec_ln_gemm(W, np.zeros(N, np.float32), g_, b_, ids=ids, pos=4, tok_emb=tok, pos_emb=pos, bf16=True)
    assert np.array_equal(xout, tok[ids] + pos[4])  # the residual stream stays fp32 in the bf16 mode
    x = (tok[ids] + pos[4]).astype(np.float64)
    ln = (x - x.mean(1, keepdims=True)) / np.sqrt(x.var(1, keepdims=True) + 1e-5)
    ref = _bf16_round(ln.astype(np.float32)).astype(np.float64) @ _bf16_round(W).astype(np.float64).T
    assert np.abs(Y - ref).max() < 1e-3


@pytest.mark.parametrize("H", [2, 6, 8])
def test_self_attention_bf16_cache_appends_and_attends(eng, H):
    """self_attention_step<true>: the prompt pass (four positions at once, causal among themselves), then one position
    at a time up to the decoder's last (30), on a bf16 cache: the appended rows ARE the bf16 roundings, every position
    reads back exactly what was appended (a wrong cache read at any position >= 5 shows here), and the new position's
    own k / v enter rounded like the cached ones."""
    rng = np.random.default_rng(30 + H)
    B, cap = 3, 32
    d = 64 * H
    kc = np.zeros((B, cap, d), np.float32)
    vc = np.zeros((B, cap, d), np.float32)
    pos = 0
    for npos in [4] + [1] * 27:
        qkv = rng.standard_normal((npos * B, 3 * d)).astype(np.float32)
        out, kc2, vc2 = eng.dbg_self_attention(qkv, kc, vc, pos, npos, bf16=True)
        for p in range(npos):
            kc[:, pos + p] = _bf16_round(qkv[p * B:(p + 1) * B, d:2 * d])
            vc[:, pos + p] = _bf16_round(qkv[p * B:(p + 1) * B, 2 * d:])
        assert np.array_equal(kc2, kc) and np.array_equal(vc2, vc), pos
        for p in range(npos):
            n = pos + p + 1
            for b in range(B):
                for h in range(H):
                    sl = slice(h * 64, (h + 1) * 64)
                    ref = attn_ref(qkv[p * B + b, sl].astype(np.float64)[None], kc[b, :n, sl].astype(np.float64),
                                   vc[b, :n, sl].astype(np.float64))[0]
                    assert np.abs(out[p * B + b, sl] - ref).max() < 1e-5, (pos, p, b, h)
        pos += npos
    assert pos == 31


@pytest.mark.parametrize("B,H,T,chunks,nq", [(2, 8, 1500, 4, 1), (3, 6, 1500, 2, 1), (2, 2, 37, 8, 1), (64, 8, 1500, 1, 2), (5, 8, 200, 2, 3),
                                             (3, 2, 64, 4, 4)])
def test_cross_attention_bf16_cache(eng, B, H, T, chunks, nq):
    """cross_attention_step<NQ, DM, true>: K / V cache rows of 64 bf16, eight lanes per key; the query projection stays
    fp32.  Against fp64 on the bf16-rounded cache: the fp32-accurate kernel's bar."""
    rng = np.random.default_rng(B + H + T + chunks + nq + 5)
    d = H * 64
    x = (rng.standard_normal((nq * B, d)) * 2 + 0.3).astype(np.float32)
    g_, b_ = (1 + 0.2 * rng.standard_normal(d)).astype(np.float32), (0.1 * rng.standard_normal(d)).astype(np.float32)
    wq = (rng.standard_normal((d, d)) / np.sqrt(d)).astype(np.float32)
    bq = (0.1 * rng.standard_normal(d)).astype(np.float32)
    kc = _bf16_round(rng.standard_normal((B, H, T, 64)).astype(np.float32))
    vc = _bf16_round(rng.standard_normal((B, H, T, 64)).astype(np.float32))
    out = eng.dbg_cross_attention(x, g_, b_, wq, bq, kc, vc, chunks, nq, bf16=True)
    q = _ln64(x, g_, b_) @ wq.astype(np.float64).T + bq
    for p in range(nq):
        for b in (range(B) if B <= 5 else (0, B // 2, B - 1)):
            for h in range(H):
                sl = slice(h * 64, (h + 1) * 64)
                ref = attn_ref(q[p * B + b, sl][None], kc[b, h].astype(np.float64), vc[b, h].astype(np.float64))[0]
                assert np.abs(out[p * B + b, sl] - ref).max() < 2e-5, (p, b, h)
