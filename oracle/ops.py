"""Oracle primitives: numpy forward + hand-derived backward of every op on the
caption-side hot path.  TEST INFRASTRUCTURE (see oracle/__init__.py).

All functions are dtype-preserving: feed float32 for the parity oracle, float64
for finite-difference checks.  Sequences are TIME-MAJOR ``[T, N, ...]`` inside
the oracle (the reference feeds batch-major ``[N, T]`` placeholders,
main.py:43-45; the transpose is pure indexing).

"TF-sem." marks behaviour of un-vendored TensorFlow 1.x / zhusuan 0.3.0 ops
restated from their public semantics; it cannot be verified in this container.
"""
import numpy as np


def sigmoid(x):
    # numerically stable logistic; tf.sigmoid (TF-sem.)
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    ex = np.exp(x[~pos])
    out[~pos] = ex / (1.0 + ex)
    return out


# --------------------------------------------------------------------------
# tf.layers.dense (activation=None): main.py:94,108; encoder.py:60-65,78-81,
# 94-97; decoder.py:111,127-129.   y = x.W + b
# --------------------------------------------------------------------------
def dense_fwd(x, W, b):
    return x @ W + b


def dense_bwd(x, W, dy):
    """returns dx, dW, db"""
    return dy @ W.T, x.T @ dy, dy.sum(axis=0)


# --------------------------------------------------------------------------
# tf.nn.embedding_lookup: encoder.py:31-36, decoder.py:77-83.
# bwd is an IndexedSlices scatter-add (TF-sem.); ``dX`` itself (one row per
# looked-up position, duplicates NOT merged) is what clip_by_global_norm sees
# (quirk Q5, ops/optimizers.py:13-16).
# --------------------------------------------------------------------------
def embedding_fwd(table, ids):
    return table[ids]


def embedding_bwd(vocab, ids, dX):
    E = dX.shape[-1]
    dense = np.zeros((vocab, E), dtype=dX.dtype)
    np.add.at(dense, ids.reshape(-1), dX.reshape(-1, E))
    return dense


# --------------------------------------------------------------------------
# tf.nn.dropout(x, keep): decoder.py:85-87; DropoutWrapper(output_keep_prob)
# utils/rnn_model.py:45-46; vgg fc dropout image_embeddings.py:225-226,236-237.
# TF-sem.: y = x * mask / keep, mask ~ Bernoulli(keep).  Masks are INJECTED.
# --------------------------------------------------------------------------
def dropout_fwd(x, mask, keep):
    return x * mask / x.dtype.type(keep)


def dropout_bwd(dy, mask, keep):
    return dy * mask / dy.dtype.type(keep)


# --------------------------------------------------------------------------
# LSTMCell step (utils/rnn_model.py:23-51 builds MultiRNNCell[DropoutWrapper(
# LSTMCell(H))]; single-step calls encoder.py:46,48, decoder.py:100,102,113).
# TF-sem. (tf.contrib.rnn.LSTMCell, no peepholes/projection/clipping):
#   g = [x, h].W + b ; i, j, f, o = split(g, 4, axis=1)
#   c' = sigmoid(f + forget_bias) * c + sigmoid(i) * tanh(j)
#   h' = sigmoid(o) * tanh(c')            forget_bias = 1.0
# W is [E+H, 4H] (rows 0..E-1 multiply x, rows E.. multiply h), b is [4H].
# --------------------------------------------------------------------------
FORGET_BIAS = 1.0


def lstm_gates_fwd(g, c_prev):
    H = c_prev.shape[1]
    i = sigmoid(g[:, 0 * H:1 * H])
    j = np.tanh(g[:, 1 * H:2 * H])
    f = sigmoid(g[:, 2 * H:3 * H] + g.dtype.type(FORGET_BIAS))
    o = sigmoid(g[:, 3 * H:4 * H])
    c = f * c_prev + i * j
    h = o * np.tanh(c)
    return i, j, f, o, c, h


def lstm_seq_fwd(X, lens_eff, W, b, c0=None, h0=None):
    """Run the one shared cell over a time-major sequence (quirk Q6: the image /
    c_v / z "init chain" steps are simply extra leading time steps of the same
    cell, encoder.py:46-55, decoder.py:100-121).

    X        [T, N, E]   inputs of every step
    lens_eff [N] int     step t of row n is active iff t < lens_eff[n]
                         (= n_init + caption length; init steps are always
                         active).  Inactive: state copied through (TF-sem.,
                         tf.nn.dynamic_rnn with sequence_length: output zero,
                         state carried; encoder.py:49-55, decoder.py:116-121).
    returns cache dict with
      cs, hs [T+1, N, H]  states (index 0 = initial), act [T, N, 4H] gate
      activations (i, j, f, o post-nonlinearity), mask [T, N]
    """
    T, N, E = X.shape
    H = W.shape[1] // 4
    dt = X.dtype
    cs = np.zeros((T + 1, N, H), dt)
    hs = np.zeros((T + 1, N, H), dt)
    if c0 is not None:
        cs[0] = c0
    if h0 is not None:
        hs[0] = h0
    act = np.zeros((T, N, 4 * H), dt)
    mask = (np.arange(T)[:, None] < np.asarray(lens_eff)[None, :])
    Wx, Wh = W[:E], W[E:]
    for t in range(T):
        g = X[t] @ Wx + hs[t] @ Wh + b
        i, j, f, o, c, h = lstm_gates_fwd(g, cs[t])
        act[t] = np.concatenate([i, j, f, o], axis=1)
        m = mask[t][:, None]
        cs[t + 1] = np.where(m, c, cs[t])
        hs[t + 1] = np.where(m, h, hs[t])
    return dict(X=X, W=W, cs=cs, hs=hs, act=act, mask=mask)


def lstm_seq_bwd(cache, dhs, dc_last=None):
    """Back-propagation through time.

    dhs [T+1, N, H]: external gradient w.r.t. every state hs[t] (index 0 = the
    initial state; normally zero).  Returns dX [T,N,E], dW, db, dc0, dh0.
    """
    X, W, cs, hs, act, mask = (cache[k] for k in ("X", "W", "cs", "hs", "act", "mask"))
    T, N, E = X.shape
    H = cs.shape[2]
    dt = X.dtype
    Wx, Wh = W[:E], W[E:]
    dG = np.zeros((T, N, 4 * H), dt)
    dh = dhs[T].copy()
    dc = np.zeros((N, H), dt) if dc_last is None else dc_last.copy()
    for t in range(T - 1, -1, -1):
        i = act[t][:, 0 * H:1 * H]
        j = act[t][:, 1 * H:2 * H]
        f = act[t][:, 2 * H:3 * H]
        o = act[t][:, 3 * H:4 * H]
        m = mask[t][:, None]
        # recompute c_t of the *active* branch (cs[t+1] equals it when active)
        tc = np.tanh(cs[t + 1])
        do = dh * tc * o * (1 - o)
        dct = dc + dh * o * (1 - tc * tc)
        di = dct * j * i * (1 - i)
        dj = dct * i * (1 - j * j)
        df = dct * cs[t] * f * (1 - f)
        g = np.concatenate([di, dj, df, do], axis=1)
        g = np.where(m, g, 0)
        dG[t] = g
        dh_prev = g @ Wh.T
        # inactive rows: state was copied through
        dh = np.where(m, dh_prev, dh) + dhs[t]
        dc = np.where(m, dct * f, dc)
    dX = dG @ Wx.T
    dWx = np.einsum("tne,tng->eg", X, dG)
    dWh = np.einsum("tnh,tng->hg", hs[:T], dG)
    dW = np.concatenate([dWx, dWh], axis=0)
    db = dG.sum(axis=(0, 1))
    return dX, dW, db, dc, dh


# --------------------------------------------------------------------------
# Reparameterised Gaussian sample.  zhusuan 0.3.0 zs.Normal(mean, std,
# n_samples=S) (encoder.py:108-109): z[s] = mean + std * eps[s], eps ~ N(0,1),
# shape [S, N, L] (TF-sem./zhusuan-sem.).  eps is INJECTED.
# --------------------------------------------------------------------------
def sample_z_fwd(mean, std, eps):
    return mean[None] + std[None] * eps


def sample_z_bwd(dz, eps):
    """returns dmean, dstd"""
    return dz.sum(axis=0), (dz * eps).sum(axis=0)


def q1_reshape(z, L, S):
    """Quirk Q1 (decoder.py:109-110): tf.reshape(z [S,N,L], [-1, L*S]) with NO
    transpose -> row r = flat[r*S*L:(r+1)*S*L]; mixes batch rows when N > 1."""
    return z.reshape(-1, L * S)


# --------------------------------------------------------------------------
# KL terms (main.py:118-145).
# --------------------------------------------------------------------------
def kl_normal_fwd(mean, std):
    """Normal and GMM priors (main.py:120-124, 131-135; quirk Q4): scalar
    -0.5 * mean_n sum_l (1 + log(std^2 + 1e-5) - mean^2 - std^2)."""
    dt = mean.dtype
    t = 1 + np.log(std * std + dt.type(1e-5)) - mean * mean - std * std
    return dt.type(-0.5) * t.sum(axis=1).mean()


def kl_normal_bwd(mean, std, dk):
    """dk = upstream gradient of the scalar KL.  returns dmean, dstd."""
    dt = mean.dtype
    N = mean.shape[0]
    s = dt.type(dk) * dt.type(-0.5) / dt.type(N)
    dmean = s * (-2 * mean)
    dstd = s * (2 * std / (std * std + dt.type(1e-5)) - 2 * std)
    return dmean, dstd


AG_SIGMA = 0.1  # utils/vae_utils.py:10  c_sigma = tf.constant(0.1)


def kl_ag_fwd(mean, std, c_i, c_means):
    """AG prior (main.py:140-145; quirk Q3): per-row VECTOR [N]
    -0.5 * sum_l [0.5 + log(std+1e-5) - log(0.1+1e-5)
                  - ((mean - c_i.Cm)^2 + std^2) / (2*0.1^2 + 1e-7)]"""
    dt = mean.dtype
    cs = dt.type(AG_SIGMA)
    mu_p = c_i @ c_means
    den = dt.type(2) * cs * cs + dt.type(1e-7)
    t = (dt.type(0.5) + np.log(std + dt.type(1e-5)) - np.log(cs + dt.type(1e-5))
         - ((mean - mu_p) ** 2 + std * std) / den)
    return dt.type(-0.5) * t.sum(axis=1)


def kl_ag_bwd(mean, std, c_i, c_means, dk_rows):
    """dk_rows [N] upstream gradient per row.  returns dmean, dstd."""
    dt = mean.dtype
    cs = dt.type(AG_SIGMA)
    mu_p = c_i @ c_means
    den = dt.type(2) * cs * cs + dt.type(1e-7)
    s = (dt.type(-0.5) * dk_rows)[:, None]
    dmean = s * (-2 * (mean - mu_p) / den)
    dstd = s * (1 / (std + dt.type(1e-5)) - 2 * std / den)
    return dmean, dstd


# --------------------------------------------------------------------------
# Masked sparse softmax cross-entropy (main.py:152-158; quirk Q8).
# TF-sem.: ce = logsumexp(logits) - logits[label] (max-shifted).
# mask = sign(label) (PAD = 0), loss = sum(ce*mask) / sum(mask).
# --------------------------------------------------------------------------
def xent_masked_fwd(logits, labels):
    mx = logits.max(axis=1, keepdims=True)
    ex = np.exp(logits - mx)
    se = ex.sum(axis=1, keepdims=True)
    lse = (np.log(se) + mx)[:, 0]
    ce = lse - logits[np.arange(logits.shape[0]), labels]
    mask = np.sign(labels).astype(logits.dtype)
    num = (ce * mask).sum()
    den = mask.sum()
    return num / den, dict(p=ex / se, labels=labels, mask=mask, den=den, num=num)


def xent_masked_bwd(cache, dloss):
    p, labels, mask, den = cache["p"], cache["labels"], cache["mask"], cache["den"]
    d = p.copy()
    d[np.arange(p.shape[0]), labels] -= 1
    return d * (mask * (p.dtype.type(dloss) / den))[:, None]
