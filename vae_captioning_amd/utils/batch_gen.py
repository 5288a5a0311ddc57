"""Batch assembly with the tensor contract of the reference's Batch_Generator
(utils/batch_gen.py:164-205, 296-345) followed by preprocess_captions
(utils/caption_utils.py:4-25): what main.py:226-238 finally feeds.

Only the caption / feature / cluster-vector side is here; JPEG / HDF5 image loading needs h5py and
cv2, which this environment lacks (images can be supplied as an in-memory array instead).
"""
import numpy as np


def form_captions_batch(indexed, names, num_captions=1, rng=None):
    """indexed: {file_name: [[ids of <BOS> w.. <EOS>], ...]}.  Returns (inputs, labels, lengths):
    inputs = caption[:-1], labels = caption[1:], lengths = len(caption) - 1; images with fewer than
    num_captions captions leave all-PAD slots of length 0; everything is padded with 0 to the longest
    caption of the batch.  num_captions == 1 picks ONE caption at random (random_select)."""
    nb = len(names)
    ins = [[[0] for _ in range(num_captions)] for _ in range(nb)]
    lab = [[[0] for _ in range(num_captions)] for _ in range(nb)]
    lengths = np.zeros((nb, num_captions), np.int32)
    for i, fn in enumerate(names):
        caps = indexed[fn.split("/")[-1]]
        if num_captions == 1 and len(caps):
            r = rng if rng is not None else np.random
            caps = [caps[int(r.integers(len(caps))) if hasattr(r, "integers") else r.randint(len(caps))]]
        for k, cap in enumerate(caps[:num_captions]):
            ins[i][k], lab[i][k], lengths[i][k] = cap[:-1], cap[1:], len(cap) - 1
    pad = max(len(c) for row in ins for c in row)
    to_arr = lambda rows: np.array([[c + [0] * (pad - len(c)) for c in row] for row in rows], np.int32)
    return to_arr(ins), to_arr(lab), lengths


def preprocess_captions(inputs, labels, lengths, c_v=None):
    """[B, nc, T] -> [B*nc, T]; cluster vectors repeated per caption (utils/caption_utils.py:4-25)."""
    B, nc, T = inputs.shape
    out = dict(cap_dec=inputs.reshape(B * nc, T), cap_enc=labels.reshape(B * nc, T), lengths=lengths.reshape(-1))
    if c_v is not None and len(c_v):
        out["c_v"] = np.repeat(np.asarray(c_v, np.float32), nc, axis=0)
    return out


class BatchGenerator(object):
    """next_batch(): dicts in the layout Trainer.set_batch takes.  features: {file_name: [1, F] or [F]}
    (the reference's feature pickles); cluster vectors: {file_name: 91-vector}, column 0 dropped
    (main.py:236)."""

    def __init__(self, indexed_captions, features, batch_size, cluster_vectors=None, seed=42):
        self.caps, self.feats, self.bs, self.cv = indexed_captions, features, batch_size, cluster_vectors
        self.names = [n for n in indexed_captions if n in features]
        self.rng = np.random.default_rng(seed)

    def next_batch(self, use_obj_vectors=False, num_captions=1, shuffle=True):
        order = self.rng.permutation(len(self.names)) if shuffle else np.arange(len(self.names))
        for s in range(0, len(order), self.bs):
            names = [self.names[i] for i in order[s:s + self.bs]]
            ins, lab, lens = form_captions_batch(self.caps, names, num_captions, self.rng)
            if ins.ndim == 2:
                ins, lab, lens = ins[:, None, :], lab[:, None, :], lens.reshape(-1, 1)
            cv = None
            if use_obj_vectors and self.cv is not None:
                cv = np.stack([np.asarray(self.cv[n], np.float32)[1:] for n in names])
            b = preprocess_captions(ins, lab, lens, cv)
            b["features"] = np.stack([np.asarray(self.feats[n], np.float32).reshape(-1) for n in names])
            b["names"] = names
            yield b
