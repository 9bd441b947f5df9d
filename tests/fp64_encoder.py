"""fp64 restatement of the encoder graph (Conv1d+GELU, Conv1d stride 2+GELU, + positional table,
L x {x += Attn(LN(x)); x += W2 GELU(W1 LN(x))}, LN) in plain numpy: the arbiter between the fp32 implementations
(CPU oracle, fp32-MFMA kernels, fp16-plane kernels) where they disagree by more than fp32 rounding — on
ill-conditioned (outlier) weights.  TEST INFRASTRUCTURE; micro-sized inputs only (dense T x T attention)."""
import numpy as np
from scipy.special import erf


def _gelu(x):
    return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))


def _ln(x, g, b):
    m = x.mean(-1, keepdims=True)
    v = ((x - m) ** 2).mean(-1, keepdims=True)
    return (x - m) / np.sqrt(v + 1e-5) * g + b


def encoder_fp64(dims, t, mel):
    """dims / t: tools/wtw.read_wtw(); mel [n_mels][2 * n_audio_ctx] -> [n_audio_ctx][d] float64"""
    f = lambda k: np.asarray(t[k], dtype=np.float64)
    d, H, L, T = dims["n_audio_state"], dims["n_audio_head"], dims["n_audio_layer"], dims["n_audio_ctx"]
    x = np.asarray(mel, dtype=np.float64)                      # [C][T0]
    T0 = x.shape[1]
    w1, b1 = f("encoder.conv1.weight"), f("encoder.conv1.bias")  # [d][C][3]
    xp = np.pad(x, ((0, 0), (1, 1)))
    h = sum(w1[:, :, k] @ xp[:, k:k + T0] for k in range(3)) + b1[:, None]
    h = _gelu(h)                                               # [d][T0]
    w2, b2 = f("encoder.conv2.weight"), f("encoder.conv2.bias")
    hp = np.pad(h, ((0, 0), (1, 1)))
    y = sum(w2[:, :, k] @ hp[:, k:k + T0:2][:, :T] for k in range(3)) + b2[:, None]
    x = _gelu(y).T + f("encoder.positional_embedding")        # [T][d]
    for l in range(L):
        p = f"encoder.blocks.{l}"
        a = _ln(x, f(p + ".attn_ln.weight"), f(p + ".attn_ln.bias"))
        q = a @ f(p + ".attn.query.weight").T + f(p + ".attn.query.bias")
        k = a @ f(p + ".attn.key.weight").T
        v = a @ f(p + ".attn.value.weight").T + f(p + ".attn.value.bias")
        o = np.empty_like(q)
        for hh in range(H):
            sl = slice(64 * hh, 64 * hh + 64)
            s = q[:, sl] @ k[:, sl].T / 8.0
            s = np.exp(s - s.max(-1, keepdims=True))
            o[:, sl] = (s / s.sum(-1, keepdims=True)) @ v[:, sl]
        x = x + o @ f(p + ".attn.out.weight").T + f(p + ".attn.out.bias")
        m = _ln(x, f(p + ".mlp_ln.weight"), f(p + ".mlp_ln.bias"))
        m = _gelu(m @ f(p + ".mlp.0.weight").T + f(p + ".mlp.0.bias"))
        x = x + m @ f(p + ".mlp.2.weight").T + f(p + ".mlp.2.bias")
    return _ln(x, f("encoder.ln_post.weight"), f("encoder.ln_post.bias"))
