#!/usr/bin/env python
"""Generates tests/golden/step_*.npz: inputs and expected outputs of 3 consecutive training
steps per prior, computed by the CPU oracle (fp64 evaluation, stored as fp32/fp64).

The reference itself cannot produce vectors (TensorFlow 1.x / zhusuan are not installable
here), so these pin the ORACLE against regressions and give the GPU
tests fixed targets; they are not outputs of the reference ("parity unpinned").
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import caption_model as cm  # noqa: E402
from oracle import decode, optim as oo  # noqa: E402
from vae_captioning_amd import spec, synth  # noqa: E402

CASES = {
    "normal": dict(prior="Normal"),
    "lstm_baseline": dict(prior="Normal", no_encoder=True),
    "ag_cv": dict(prior="AG", use_c_v=True),
    "gmm": dict(prior="GMM"),
}
DIMS = dict(embed_size=8, encoder_hidden=32, decoder_hidden=32, latent_size=6, gen_z_samples=3, num_captions=2,
            cnn_feature_size=16, vocab_size=50)
CLIP, LR, NSTEPS = 0.05, 5e-4, 3


def chk(a):
    """[sum, L2 norm] checksum: the 360 GMM/AG head tensors are stored as checksums to keep the fixture small."""
    a = np.asarray(a, np.float64)
    return np.array([a.sum(), np.sqrt((a * a).sum())], np.float64)


def pack_heads(out):
    """The 360 head tensors (and their checksums) as 4 stacked arrays in sorted-name order
    (zip per-entry overhead would otherwise dominate the file)."""
    res = {k: v for k, v in out.items() if "_ll_" not in k}
    for grp in ("p/", "g/", "q/"):
        ks = sorted(k for k in out if k.startswith(grp) and "_ll_" in k and k.endswith("kernel"))
        bs = sorted(k for k in out if k.startswith(grp) and "_ll_" in k and k.endswith("bias"))
        if ks:
            res["heads_kernel_" + grp[0]] = np.stack([out[k] for k in ks])
            res["heads_bias_" + grp[0]] = np.stack([out[k] for k in bs])
    return res


def run_case(name, kw):
    cfg = cm.default_cfg(**DIMS, **kw)
    V = cfg.vocab_size
    rng = np.random.default_rng(sum(map(ord, name)))
    P = spec.init_caption_params(cfg, V, seed=11)
    for k in P:
        if k.endswith("bias"):
            P[k] = rng.normal(0, 0.1, P[k].shape).astype(np.float32)
    batch = synth.make_batch(rng, 2, cfg.num_captions, 5, V, use_ci=spec.uses_ci(cfg), variable_len=True, feature_size=cfg.cnn_feature_size)
    noise = synth.make_noise(rng, 4, 5, cfg)
    out = {"p/" + k: v for k, v in P.items()}
    out.update({"b/" + k: v for k, v in batch.items()})
    out.update({"n/" + k: v for k, v in noise.items()})
    f64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
    P64, b64, n64 = f64(P), f64(batch), f64(noise)
    if cfg.prior == "AG":
        n64["c_means"] = decode.init_clusters(90, cfg.latent_size).astype(np.float64)
    st = {}
    for s in range(NSTEPS):
        r = cm.forward_backward(P64, b64, n64, cfg, global_step=s)
        norm = float(oo.global_norm({k: v.astype(np.float32) for k, v in r.grads.items()}, {k: v.astype(np.float32) for k, v in r.sparse.items()}))
        scale = CLIP * min(1.0 / norm, 1.0 / CLIP)
        out["e/step%d" % s] = np.array([float(np.mean(r.kld)), float(r.rec_loss), float(np.mean(r.lower_bound)), norm], np.float64)
        if s == 0:
            for k, g in r.grads.items():
                out["g/" + k] = chk(g) if "_ll_" in k else g.astype(np.float32)
        P32 = {k: v.astype(np.float32) for k, v in P64.items()}
        oo.adam_step(P32, {k: v.astype(np.float32) for k, v in r.grads.items()}, st, LR, s + 1, scale=scale)
        P64 = f64(P32)
    for k, v in P64.items():  # parameters after NSTEPS steps
        out["q/" + k] = chk(v) if "_ll_" in k else v.astype(np.float32)
    out = pack_heads(out)
    np.savez_compressed(os.path.join(HERE, "step_%s.npz" % name), **out)
    print(name, {k: out[k] for k in out if k.startswith("e/")})


if __name__ == "__main__":
    for name, kw in CASES.items():
        run_case(name, kw)
    cmn = decode.init_clusters(90, 150)
    np.savez_compressed(os.path.join(HERE, "cluster_means_seed42.npz"), c_means=cmn)
    # The reference ships one data file on this path: obj_vectors/category_index.pickle, the MSCOCO category table
    # ({id: {'id', 'name'}}, 80 entries) whose id gaps are decoder.py:56's `un_clusters`.  Stored as data (json), only
    # when the reference tree is mounted (it is not on the GPU box).
    ref = "/root/reference/obj_vectors/category_index.pickle"
    if os.path.exists(ref):
        import json
        import pickle
        with open(ref, "rb") as fh:
            cat = pickle.load(fh)
        with open(os.path.join(HERE, "category_index.json"), "w") as fh:
            json.dump({str(k): cat[k]["name"] for k in sorted(cat)}, fh, indent=0)
