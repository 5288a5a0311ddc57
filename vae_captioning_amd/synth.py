"""Seeded synthetic batches with the tensor contract of the reference's batch
generator (utils/batch_gen.py:164-205,296-345 + utils/caption_utils.py:4-25;
SURVEY.md section 8 rows a14, d).  There is no MSCOCO here, so shapes and value
ranges follow the reference, contents are random.
"""
import numpy as np

BOS, EOS = 1, 2  # arbitrary synthetic ids (real ids are frequency-ranked)


def make_batch(rng, B, nc, T, vocab, *, use_ci=False, images=False, variable_len=False,
               feature_size=4096):
    """Returns dict: features [B,F] f32 (or images [B,224,224,3] f32 0..255),
    cap_dec [N,T] i32 ("<BOS> w.."), cap_enc [N,T] i32 ("w.. <EOS>"),
    lengths [N] i32 (= tokens - 1), c_v [N,90] f32 (sum-normalised indicators)."""
    N = B * nc
    if variable_len:
        lens = np.clip(np.rint(rng.normal(11, 3, size=N)), 6, T).astype(np.int32)
        lens[rng.random(N) < 0.02] = 0  # images with < nc captions give all-PAD rows (batch_gen.py:313-317)
        if T >= 6:
            lens[0] = T
    else:
        lens = np.full(N, T, np.int32)
    words = rng.integers(3, vocab, size=(N, T + 1), dtype=np.int64).astype(np.int32)
    cap_dec = np.zeros((N, T), np.int32)
    cap_enc = np.zeros((N, T), np.int32)
    for n in range(N):
        l = int(lens[n])
        if l == 0:
            continue
        seq = np.concatenate([[BOS], words[n, :l - 1], [EOS]]).astype(np.int32)  # l+1 tokens
        cap_dec[n, :l] = seq[:-1]
        cap_enc[n, :l] = seq[1:]
    out = dict(cap_dec=cap_dec, cap_enc=cap_enc, lengths=lens)
    if images:
        out["images"] = rng.integers(0, 256, size=(B, 224, 224, 3), dtype=np.uint8)
        if images != "u8":   # images="u8": the pixels as the reference's HDF5 holds them (preprocess.py:27-28); default: the float32 feed
            out["images"] = out["images"].astype(np.float32)
    else:
        out["features"] = np.maximum(rng.standard_normal((B, feature_size), dtype=np.float32), 0)
    if use_ci:
        cv = np.zeros((B, 90), np.float32)
        for b in range(B):
            k = int(rng.integers(1, 6))
            cv[b, rng.choice(90, size=k, replace=False)] = 1.0
        cv /= cv.sum(axis=1, keepdims=True)
        out["c_v"] = np.repeat(cv, nc, axis=0)  # caption_utils.py:21-23
    return out


def make_noise(rng, N, T, p, *, Hd=None):
    """Injected randomness: eps [S,N,L], optional dropout masks, GMM indices."""
    out = {}
    if not p.no_encoder:
        out["eps"] = rng.standard_normal((p.gen_z_samples, N, p.latent_size), dtype=np.float32)
        if p.prior == "GMM":
            out["gmm_idx"] = rng.integers(0, 90, size=N).astype(np.int32)
    if p.dec_keep_rate < 1:
        out["drop_in"] = (rng.random((T, N, p.embed_size)) < p.dec_keep_rate).astype(np.float32)
    if p.dec_lstm_drop < 1:
        out["drop_out"] = (rng.random((T, N, Hd or p.decoder_hidden)) < p.dec_lstm_drop).astype(np.float32)
    return out
