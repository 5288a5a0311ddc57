"""Oracle: caption generation (vae_model/decoder.py:145-320, utils/top_n.py) and
cluster-mean initialisation (utils/vae_utils.py).  TEST INFRASTRUCTURE.
"""
import heapq

import numpy as np

from . import ops
from .caption_model import DEC_CELL, uses_ci

# decoder.py:56 -- category ids absent from obj_vectors/category_index.pickle
UN_CLUSTERS = {0, 66, 68, 69, 71, 12, 45, 83, 26, 29, 30}


def init_clusters(num_clusters=90, latent_size=150, seed=42):
    """utils/vae_utils.py:20-27 with the numpy global RNG seeded 42 by
    Batch_Generator.__init__ (utils/batch_gen.py:65-66).  Known answer:
    Cm[0,:5] = [-0.03451613, 0.1239991, 0.06382544, 0.02714261, -0.09463507]."""
    rs = np.random.RandomState(seed)
    rows = []
    for _ in range(num_clusters):
        item = 2 * rs.random_sample((1, latent_size)) - 1
        item = item / np.sqrt(np.sum(item ** 2))
        rows.append(item)
    return np.squeeze(np.stack(rows).astype(np.float32))


class TopN(object):
    """utils/top_n.py:4-43 (bounded min-heap; ordering by Beam.score only)."""

    def __init__(self, n):
        self._n = n
        self._data = []

    def size(self):
        return len(self._data)

    def push(self, x):
        if len(self._data) < self._n:
            heapq.heappush(self._data, x)
        else:
            heapq.heappushpop(self._data, x)

    def extract(self, sort=False):
        data = self._data
        self._data = None
        if sort:
            data.sort(reverse=True)
        return data

    def reset(self):
        self._data = []


class Beam(object):
    """utils/top_n.py:46-72."""

    def __init__(self, sentence, state, logprob, score):
        self.sentence = sentence
        self.logprob = logprob
        self.state = state
        self.score = score

    def __lt__(self, other):
        return self.score < other.score

    def __eq__(self, other):
        return self.score == other.score


def gen_prior_mean(cfg, c_v_row, c_means):
    """decoder.py:42-71: zero mean, or (AG, gen mode) the mean of the image's
    cluster means; empty cluster vector -> ids not in UN_CLUSTERS (quirk Q16:
    raw ids up to 90 index a 90-row matrix; id 90 is clamped here because
    tf.gather on GPU returns zeros / CPU raises -- flagged, not reproducible)."""
    L = cfg.latent_size
    if cfg.prior != "AG":
        return np.zeros((1, L), np.float32)
    idx = np.nonzero(c_v_row > 0)[0]
    if idx.size == 0:
        idx = np.array([i for i in range(cfg.num_clusters + 1) if i not in UN_CLUSTERS and i < c_means.shape[0]])
    return c_means[idx].mean(axis=0).reshape(1, L).astype(np.float32)


def initial_state(P, cfg, feature, c_v_row, eps, c_means=None, std=0.1):
    """State after the init chain image -> (c_v) -> z  (decoder.py:96-114) for ONE
    image (batch 1, so the Q1 reshape is the identity on [S,1,L] -> [1,S*L])."""
    images_fv = ops.dense_fwd(feature[None], P["imf_emb/kernel"], P["imf_emb/bias"])
    xs = [images_fv]
    if cfg.use_c_v and uses_ci(cfg):
        xs.append(ops.dense_fwd(c_v_row[None], P["cv_emb/kernel"], P["cv_emb/bias"]))
    if not cfg.no_encoder:
        mean = gen_prior_mean(cfg, c_v_row, c_means)
        z = mean[None] + np.float32(std) * eps  # [S,1,L]
        zin = z.reshape(1, -1)
        xs.append(ops.dense_fwd(zin, P["decoder/net/z_rnn/kernel"], P["decoder/net/z_rnn/bias"]))
    X = np.stack(xs, axis=0)
    c = ops.lstm_seq_fwd(X, np.array([len(xs)]), P[DEC_CELL + "kernel"], P[DEC_CELL + "bias"])
    return c["cs"][-1], c["hs"][-1]


def step(P, token, state):
    """One gen-mode decoder step: returns softmax probs [V] and new state."""
    c, h = state
    x = P["decoder/net/dec_embeddings"][np.array([token])][None]  # [1,1,E]
    r = ops.lstm_seq_fwd(x, np.array([1]), P[DEC_CELL + "kernel"], P[DEC_CELL + "bias"], c, h)
    hn = r["hs"][-1]
    logits = ops.dense_fwd(hn, P["decoder/rnn_logits/kernel"], P["decoder/rnn_logits/bias"])[0]
    e = np.exp(logits - logits.max())
    return e / e.sum(), (r["cs"][-1], hn)


def greedy(P, cfg, feature, c_v_row, eps, bos, eos, c_means=None, max_len=30):
    """decoder.py:145-201 (online_inference, sample_gen='greedy')."""
    state = initial_state(P, cfg, feature, c_v_row, eps, c_means, std=getattr(cfg, "std", 0.1))
    tok = bos
    out = []
    for _ in range(max_len):
        probs, state = step(P, tok, state)
        tok = int(np.argmax(probs))  # p**(1/t)/sum is argmax-invariant (decoder.py:184-189)
        out.append(tok)
        if tok == eos:
            break
    return out


def beam_search(P, cfg, feature, c_v_row, eps, bos, eos, c_means=None, beam_size=2,
                max_len=30, len_norm_f=0.7):
    """decoder.py:203-320.  <BOS> is consumed twice."""
    state = initial_state(P, cfg, feature, c_v_row, eps, c_means, std=getattr(cfg, "std", 0.1))
    _, state = step(P, bos, state)  # decoder.py:230-236, probs discarded
    partial = TopN(beam_size)
    partial.push(Beam([bos], state, 0.0, 0.0))
    complete = TopN(beam_size)
    for _ in range(max_len - 1):
        plist = partial.extract()
        partial.reset()
        res = [step(P, c.sentence[-1], c.state) for c in plist]
        for pc, (probs, st) in zip(plist, res):
            w_probs = list(enumerate(probs.ravel()))
            w_probs.sort(key=lambda x: -x[1])  # stable: ties -> lower index first
            for w, p in w_probs[:beam_size]:
                if p < 1e-12:
                    continue
                sentence = pc.sentence + [w]
                logprob = pc.logprob + float(np.log(np.float32(p)))  # decoder.py:282: float32 log (p is a float32 softmax output), float64 sum
                score = logprob
                if w == eos:
                    if len_norm_f > 0:
                        score /= len(sentence) ** len_norm_f
                    complete.push(Beam(sentence, st, logprob, score))
                else:
                    partial.push(Beam(sentence, st, logprob, score))
        if partial.size() == 0:
            break
    if not complete.size():
        complete = partial
    beams = complete.extract(sort=True)
    return [b.sentence for b in beams], [b.score for b in beams]
