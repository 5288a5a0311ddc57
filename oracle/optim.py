"""Oracle: ops/optimizers.py restated.  TEST INFRASTRUCTURE (oracle/__init__.py).

TF-sem. (un-vendored tf.train.* / tf.clip_by_global_norm, TF 1.4), unverifiable here:
  * clip_by_global_norm: norm = sqrt(sum_t ||t||^2) with IndexedSlices
    contributing their un-deduplicated ``values`` (quirk Q5);
    scale = clip * min(1/norm, 1/clip).
  * AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1*m+(1-b1)*g;
    v = b2*v+(1-b2)*g^2; var -= lr_t*m/(sqrt(v)+eps).  Sparse gradients are
    de-duplicated (summed) first and m, v decay densely, so the embedding
    update equals dense Adam on the scatter-added gradient.
  * MomentumOptimizer: accum = mom*accum + g; var -= lr*accum; the sparse
    variant touches only the rows present in the batch.
  * exponential_decay(staircase): lr * rate^floor(step/decay_steps)
    (ops/optimizers.py:24-31, quirk Q13: Adam ignores it).
"""
import numpy as np

EMBEDDINGS = ("encoder/enc_embeddings", "decoder/net/dec_embeddings")


def global_norm(grads, sparse):
    """ops/optimizers.py:15-16 (Q5).  float64 accumulate of float32 squares."""
    tot = 0.0
    for n, g in grads.items():
        if n in sparse:
            v = sparse[n]
            tot += float(np.sum(v.astype(np.float64) ** 2))
        else:
            tot += float(np.sum(g.astype(np.float64) ** 2))
    return np.float32(np.sqrt(tot))


def clip_scale(norm, clip):
    norm = np.float32(norm)
    clip = np.float32(clip)
    return np.float32(clip * min(np.float32(1) / norm, np.float32(1) / clip)) if norm > 0 else np.float32(1)


def decayed_lr(lr, global_step, num_ex_per_epoch=150000, batch_size=32, num_epochs_per_decay=5):
    decay_steps = int(num_ex_per_epoch / (batch_size + 0.001) * num_epochs_per_decay)
    return np.float32(lr) * np.float32(0.5) ** np.float32(global_step // decay_steps)


def adam_step(P, grads, state, lr, t, beta1=0.8, beta2=0.999, eps=1e-8, scale=1.0, l2=0.0):
    """In-place.  t = 1-based step count.  ``scale`` = clip scale (1 for the CNN
    optimiser, ops/optimizers.py:49-82).  ``l2``: adds l2*w to the gradient
    (gradient of l2_regularizer(l2)(w) = l2*sum(w^2)/2, main.py:69-74)."""
    f = np.float32
    lr_t = f(f(lr) * np.sqrt(f(1) - f(beta2) ** f(t)) / (f(1) - f(beta1) ** f(t)))
    for n, g in grads.items():
        g = g * f(scale)
        if l2:
            g = g + f(l2) * P[n]
        m = state.setdefault("m/" + n, np.zeros_like(P[n]))
        v = state.setdefault("v/" + n, np.zeros_like(P[n]))
        m[...] = f(beta1) * m + f(1 - beta1) * g
        v[...] = f(beta2) * v + f(1 - beta2) * g * g
        P[n] -= lr_t * m / (np.sqrt(v) + f(eps))


def sgd_step(P, grads, lr, scale=1.0, l2=0.0):
    f = np.float32
    for n, g in grads.items():
        g = g * f(scale)
        if l2:
            g = g + f(l2) * P[n]
        P[n] -= f(lr) * g


def momentum_step(P, grads, state, lr, momentum=0.9, scale=1.0, touched=None, l2=0.0):
    """touched: {embedding name -> bool [V]} rows present in this batch's indices."""
    f = np.float32
    for n, g in grads.items():
        g = g * f(scale)
        if l2:
            g = g + f(l2) * P[n]
        a = state.setdefault("a/" + n, np.zeros_like(P[n]))
        if touched is not None and n in touched:
            r = touched[n]
            a[r] = f(momentum) * a[r] + g[r]
            P[n][r] -= f(lr) * a[r]
        else:
            a[...] = f(momentum) * a + g
            P[n] -= f(lr) * a
