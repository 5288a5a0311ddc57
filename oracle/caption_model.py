"""Oracle: the caption-side training graph (loss, KL, every gradient) restated
from main.py:84-177, vae_model/encoder.py:24-110, vae_model/decoder.py:34-143.
TEST INFRASTRUCTURE (see oracle/__init__.py).  Parameters are a plain dict keyed
by the reference's checkpoint variable names.
"""
from types import SimpleNamespace

import numpy as np

from . import ops

NUM_CLUSTERS = 90  # utils/parameters.py:4; encoder.py:76,92 hard-code range(90)

ENC_CELL = "encoder/multi_rnn_cell/cell_0/lstm_cell/"
DEC_CELL = "decoder/net/multi_rnn_cell/cell_0/lstm_cell/"


def default_cfg(**kw):
    """Defaults = utils/parameters.py:2-66."""
    c = dict(prior="Normal", no_encoder=False, use_c_v=False, num_captions=5,
             embed_size=256, encoder_hidden=512, decoder_hidden=512,
             latent_size=150, gen_z_samples=100, vocab_size=None,
             dec_keep_rate=1.0, dec_lstm_drop=1.0, ann_param=0.0,
             fine_tune=False, restore=False, mode="training",
             cnn_feature_size=4096)
    c.update(kw)
    return SimpleNamespace(**c)


def uses_ci(cfg):
    # main.py:52-53,103-104
    return cfg.use_c_v or cfg.prior in ("GMM", "AG")


def head_names(cfg, k):
    scope = "encoder/gmm_ll_%d/" % k if cfg.prior == "GMM" else "encoder/ag_ll_%d/" % k
    return scope + "dense/", scope + "dense_1/"


def annealing(cfg, global_step):
    """main.py:162-170 (quirk Q10)."""
    if cfg.fine_tune or cfg.restore:
        return 1.0
    if cfg.ann_param > 1:
        return float((np.tanh((np.float32(global_step) - 1000 * cfg.ann_param) / 1000) + 1) / 2)
    return 1.0


def _stack_heads(P, cfg):
    Wm = np.stack([P[head_names(cfg, k)[0] + "kernel"] for k in range(NUM_CLUSTERS)])
    bm = np.stack([P[head_names(cfg, k)[0] + "bias"] for k in range(NUM_CLUSTERS)])
    Ws = np.stack([P[head_names(cfg, k)[1] + "kernel"] for k in range(NUM_CLUSTERS)])
    bs = np.stack([P[head_names(cfg, k)[1] + "bias"] for k in range(NUM_CLUSTERS)])
    return Wm, bm, Ws, bs


def forward_backward(P, batch, noise, cfg, global_step=0, reg_loss=0.0, want_grads=True, q1_groups=1, dp=None):
    """One training-step evaluation.

    batch: features [B, F] f32 (fc2 features, precomputed or from the VGG oracle)
           cap_dec  [N, T] i32  "<BOS> w.."  (ann_inputs_dec, main.py:231)
           cap_enc  [N, T] i32  "w.. <EOS>"  (ann_inputs_enc = labels, main.py:230,152)
           lengths  [N]    i32
           c_v      [N, 90] f32 (only when uses_ci)
    noise: eps [S, N, L]; gmm_idx [N] (GMM); drop_in [T, N, E] / drop_out
           [T, N, H] Bernoulli masks when the keep rates are < 1.
    q1_groups: the Q1 reshape mixes rows inside each of `q1_groups` contiguous row groups
           (1 = the reference; G = what G data-parallel towers of the reference graph compute).
    dp:    None, or dict(ce_den=global non-PAD count, n_rows=global row count): evaluate ONE
           data-parallel shard so that summing the returned gradients / ce_num / kl_sum over
           shards gives the global-batch result (DESIGN.md section "data parallelism").
    returns SimpleNamespace(kld, rec_loss, lower_bound, ann, grads, sparse, dfeatures, aux)
    """
    dt = batch["features"].dtype
    nc = cfg.num_captions if cfg.mode == "training" else 1
    feats = batch["features"]
    B = feats.shape[0]
    # main.py:84-89: row i -> rows i*nc .. i*nc+nc-1
    feats_t = np.repeat(feats, nc, axis=0) if nc > 1 else feats
    N = feats_t.shape[0]
    cap_dec_t = np.ascontiguousarray(batch["cap_dec"].T)  # [T, N]
    cap_enc_t = np.ascontiguousarray(batch["cap_enc"].T)
    T = cap_dec_t.shape[0]
    lengths = np.asarray(batch["lengths"])
    E, L, S = cfg.embed_size, cfg.latent_size, cfg.gen_z_samples
    V = P["decoder/rnn_logits/kernel"].shape[1]

    images_fv = ops.dense_fwd(feats_t, P["imf_emb/kernel"], P["imf_emb/bias"])  # main.py:94
    ci = None
    ci_emb = None
    if uses_ci(cfg):
        ci = batch["c_v"]
        ci_emb = ops.dense_fwd(ci, P["cv_emb/kernel"], P["cv_emb/bias"])  # main.py:108
    feed_cv = cfg.use_c_v and ci_emb is not None  # encoder.py:47, decoder.py:101

    enc = None
    kld = dt.type(0.0)
    if not cfg.no_encoder:
        # ---- encoder.py:24-110 ----
        n_init_e = 1 + int(feed_cv)
        xs = [images_fv[None]]
        if feed_cv:
            xs.append(ci_emb[None])
        xw_e = ops.embedding_fwd(P["encoder/enc_embeddings"], cap_enc_t)  # Q7: labels
        xs.append(xw_e)
        Xe = np.concatenate(xs, axis=0)
        ce = ops.lstm_seq_fwd(Xe, n_init_e + lengths, P[ENC_CELL + "kernel"], P[ENC_CELL + "bias"])
        hT = ce["hs"][-1]  # encoder.py:58  final_state[0][1] = h
        enc = SimpleNamespace(n_init=n_init_e, cache=ce, hT=hT)
        if cfg.prior == "Normal":
            mean = ops.dense_fwd(hT, P["encoder/dense/kernel"], P["encoder/dense/bias"])
            logstd = ops.dense_fwd(hT, P["encoder/dense_1/kernel"], P["encoder/dense_1/bias"])
            std = np.exp(logstd)
        else:
            Wm, bm, Ws, bs = _stack_heads(P, cfg)
            tm = np.einsum("nh,khl->nkl", hT, Wm) + bm[None]  # [N, 90, L]
            tl = np.einsum("nh,khl->nkl", hT, Ws) + bs[None]
            etl = np.exp(tl)
            enc.tm, enc.tl, enc.etl = tm, tl, etl
            if cfg.prior == "GMM":
                k = np.asarray(noise["gmm_idx"])  # encoder.py:72-75 (Q15), injected
                mean = tm[np.arange(N), k]        # encoder.py:87-88
                std = etl[np.arange(N), k]
            else:  # AG, encoder.py:105-107
                mean = np.einsum("nk,nkl->nl", ci, tm)
                std = np.einsum("nk,nkl->nl", ci, etl)
        eps = noise["eps"]
        z = ops.sample_z_fwd(mean, std, eps)  # [S, N, L]
        enc.mean, enc.std, enc.z = mean, std, z
        # ---- KL, main.py:118-145 ----
        if cfg.prior in ("Normal", "GMM"):
            kld = ops.kl_normal_fwd(mean, std)
            if dp is not None:  # this shard's share of the global batch mean
                kld = kld * dt.type(N) / dt.type(dp["n_rows"])
        else:
            kld = ops.kl_ag_fwd(mean, std, ci, noise["c_means"])

    # ---- decoder.py:34-143 ----
    n_init_d = 1 + int(feed_cv) + int(not cfg.no_encoder)
    xs = [images_fv[None]]
    if feed_cv:
        xs.append(ci_emb[None])
    if not cfg.no_encoder:
        if q1_groups == 1:
            zin = ops.q1_reshape(enc.z, L, S)  # decoder.py:109-110 (Q1)
        else:
            ng = N // q1_groups
            zin = np.concatenate([ops.q1_reshape(enc.z[:, g * ng:(g + 1) * ng], L, S) for g in range(q1_groups)], axis=0)
        z_dec = ops.dense_fwd(zin, P["decoder/net/z_rnn/kernel"], P["decoder/net/z_rnn/bias"])
        xs.append(z_dec[None])
    xw_d = ops.embedding_fwd(P["decoder/net/dec_embeddings"], cap_dec_t)
    if cfg.dec_keep_rate < 1:
        xw_d = ops.dropout_fwd(xw_d, noise["drop_in"], cfg.dec_keep_rate)  # decoder.py:85-87
    xs.append(xw_d)
    Xd = np.concatenate(xs, axis=0)
    cd = ops.lstm_seq_fwd(Xd, n_init_d + lengths, P[DEC_CELL + "kernel"], P[DEC_CELL + "bias"])
    word_mask = cd["mask"][n_init_d:]  # [T, N]
    outs = np.where(word_mask[:, :, None], cd["hs"][n_init_d + 1:], 0)  # dynamic_rnn: zero output past length
    if cfg.dec_lstm_drop < 1:
        outs = ops.dropout_fwd(outs, noise["drop_out"], cfg.dec_lstm_drop)  # rnn_model.py:45-46
    Hd = outs.shape[2]
    outs_r = outs.reshape(T * N, Hd)
    logits = ops.dense_fwd(outs_r, P["decoder/rnn_logits/kernel"], P["decoder/rnn_logits/bias"])
    labels = cap_enc_t.reshape(-1)  # main.py:152 (time-major permutation of the same multiset)
    ce_loss, xc = ops.xent_masked_fwd(logits, labels)
    if dp is not None:
        xc["den"] = dt.type(dp["ce_den"])
        ce_loss = xc["num"] / xc["den"]
    rec_loss = ce_loss + dt.type(reg_loss)  # main.py:159-160 (Q9)
    ann = dt.type(annealing(cfg, global_step))
    if cfg.no_encoder:
        lower_bound = rec_loss  # main.py:175-177
        kld = dt.type(0.0)
    else:
        lower_bound = rec_loss + ann * kld / dt.type(10)  # main.py:173-174
    out = SimpleNamespace(kld=kld, rec_loss=rec_loss, lower_bound=lower_bound, ann=ann,
                          ce_num=xc["num"], ce_den=xc["den"],
                          aux=SimpleNamespace(logits=logits, images_fv=images_fv, enc=enc,
                                              dec_hs=cd["hs"], n_init_d=n_init_d))
    if not want_grads:
        return out

    # ================= backward of sum(lower_bound) (ops/optimizers.py:13) =======
    vector_loss = (not cfg.no_encoder) and cfg.prior == "AG"  # Q3
    n_rows = N if dp is None else dp["n_rows"]
    d_rec = dt.type(n_rows) if vector_loss else dt.type(1)
    G = {}
    sparse = {}
    dlogits = ops.xent_masked_bwd(xc, d_rec)
    douts_r, G["decoder/rnn_logits/kernel"], G["decoder/rnn_logits/bias"] = ops.dense_bwd(
        outs_r, P["decoder/rnn_logits/kernel"], dlogits)
    douts = douts_r.reshape(T, N, Hd)
    if cfg.dec_lstm_drop < 1:
        douts = ops.dropout_bwd(douts, noise["drop_out"], cfg.dec_lstm_drop)
    douts = np.where(word_mask[:, :, None], douts, 0)
    dhs = np.zeros_like(cd["hs"])
    dhs[n_init_d + 1:] = douts
    dXd, G[DEC_CELL + "kernel"], G[DEC_CELL + "bias"], _, _ = ops.lstm_seq_bwd(cd, dhs)
    d_images_fv = dXd[0].copy()
    d_ci_emb = None
    idx = 1
    if feed_cv:
        d_ci_emb = dXd[idx].copy()
        idx += 1
    if not cfg.no_encoder:
        dz_dec = dXd[idx]
        idx += 1
    dxw_d = dXd[idx:]
    if cfg.dec_keep_rate < 1:
        dxw_d = ops.dropout_bwd(dxw_d, noise["drop_in"], cfg.dec_keep_rate)
    sparse["decoder/net/dec_embeddings"] = dxw_d.reshape(-1, E)  # IndexedSlices.values (Q5)
    G["decoder/net/dec_embeddings"] = ops.embedding_bwd(V, cap_dec_t, dxw_d)

    if not cfg.no_encoder:
        dzin, G["decoder/net/z_rnn/kernel"], G["decoder/net/z_rnn/bias"] = ops.dense_bwd(
            zin, P["decoder/net/z_rnn/kernel"], dz_dec)
        if q1_groups == 1:
            dz = dzin.reshape(S, N, L)
        else:
            ng = N // q1_groups
            dz = np.concatenate([dzin[g * ng:(g + 1) * ng].reshape(S, ng, L) for g in range(q1_groups)], axis=1)
        dmean, dstd = ops.sample_z_bwd(dz, noise["eps"])
        if vector_loss:
            dk = np.full((N,), ann / dt.type(10), dt)
            km, ks = ops.kl_ag_bwd(enc.mean, enc.std, ci, noise["c_means"], dk)
        else:
            km, ks = ops.kl_normal_bwd(enc.mean, enc.std, ann / dt.type(10) * dt.type(N) / dt.type(n_rows))
        dmean = dmean + km
        dstd = dstd + ks
        hT = enc.hT
        if cfg.prior == "Normal":
            dlogstd = dstd * enc.std
            dh1, G["encoder/dense/kernel"], G["encoder/dense/bias"] = ops.dense_bwd(
                hT, P["encoder/dense/kernel"], dmean)
            dh2, G["encoder/dense_1/kernel"], G["encoder/dense_1/bias"] = ops.dense_bwd(
                hT, P["encoder/dense_1/kernel"], dlogstd)
            dhT = dh1 + dh2
        else:
            Wm, bm, Ws, bs = _stack_heads(P, cfg)
            if cfg.prior == "GMM":
                k = np.asarray(noise["gmm_idx"])
                dtm = np.zeros_like(enc.tm)
                dtl = np.zeros_like(enc.tl)
                dtm[np.arange(N), k] = dmean
                dtl[np.arange(N), k] = dstd * enc.etl[np.arange(N), k]
            else:
                dtm = ci[:, :, None] * dmean[:, None, :]
                dtl = ci[:, :, None] * dstd[:, None, :] * enc.etl
            dhT = np.einsum("nkl,khl->nh", dtm, Wm) + np.einsum("nkl,khl->nh", dtl, Ws)
            dWm = np.einsum("nh,nkl->khl", hT, dtm)
            dWs = np.einsum("nh,nkl->khl", hT, dtl)
            dbm = dtm.sum(axis=0)
            dbs = dtl.sum(axis=0)
            for kk in range(NUM_CLUSTERS):
                a, b = head_names(cfg, kk)
                G[a + "kernel"], G[a + "bias"] = dWm[kk], dbm[kk]
                G[b + "kernel"], G[b + "bias"] = dWs[kk], dbs[kk]
        ce = enc.cache
        dhs_e = np.zeros_like(ce["hs"])
        dhs_e[-1] = dhT
        dXe, G[ENC_CELL + "kernel"], G[ENC_CELL + "bias"], _, _ = ops.lstm_seq_bwd(ce, dhs_e)
        d_images_fv += dXe[0]
        idx = 1
        if feed_cv:
            d_ci_emb += dXe[idx]
            idx += 1
        dxw_e = dXe[idx:]
        sparse["encoder/enc_embeddings"] = dxw_e.reshape(-1, E)
        G["encoder/enc_embeddings"] = ops.embedding_bwd(V, cap_enc_t, dxw_e)

    dfeats_t, G["imf_emb/kernel"], G["imf_emb/bias"] = ops.dense_bwd(feats_t, P["imf_emb/kernel"], d_images_fv)
    if uses_ci(cfg):
        if d_ci_emb is not None:
            _, G["cv_emb/kernel"], G["cv_emb/bias"] = ops.dense_bwd(ci, P["cv_emb/kernel"], d_ci_emb)
        else:  # variable exists but is not on the loss path -> tf.gradients gives None
            G["cv_emb/kernel"] = np.zeros_like(P["cv_emb/kernel"])
            G["cv_emb/bias"] = np.zeros_like(P["cv_emb/bias"])
    out.grads = G
    out.sparse = sparse
    out.dfeatures = dfeats_t.reshape(B, nc, -1).sum(axis=1) if nc > 1 else dfeats_t
    return out


def trainable_names(cfg, P):
    """ops/optimizers.py:4-12: cv_emb + imf_emb + decoder/* (+ encoder/*)."""
    names = [n for n in P if n.startswith("cv_emb/")]
    names += [n for n in P if n.startswith("imf_emb/")]
    names += [n for n in P if n.startswith("decoder/")]
    if not cfg.no_encoder:
        names += [n for n in P if n.startswith("encoder/")]
    return names
