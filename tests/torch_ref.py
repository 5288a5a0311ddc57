"""Independent torch-CPU statement of the reference training graph, used ONLY to
cross-check the oracle's hand-derived backward passes through autograd
(SURVEY.md section 8c "independent cross-checks").  Written against the
reference sources directly (main.py:84-177, encoder.py, decoder.py), batch-major
like the reference, sharing no code with oracle/.
"""
import torch

ENC = "encoder/multi_rnn_cell/cell_0/lstm_cell/"
DEC = "decoder/net/multi_rnn_cell/cell_0/lstm_cell/"


def cell(x, c, h, W, b):
    g = torch.cat([x, h], 1) @ W + b
    i, j, f, o = torch.chunk(g, 4, dim=1)
    c2 = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return c2, h2


def dynamic_rnn(x_bt, lengths, c, h, W, b):
    """x_bt [N, T, E]; returns outputs [N, T, H] (zero past length) and final state."""
    outs = []
    for t in range(x_bt.shape[1]):
        c2, h2 = cell(x_bt[:, t], c, h, W, b)
        m = (t < lengths).unsqueeze(1)
        outs.append(torch.where(m, h2, torch.zeros_like(h2)))
        c = torch.where(m, c2, c)
        h = torch.where(m, h2, h)
    return torch.stack(outs, 1), (c, h)


def forward(P, batch, noise, cfg, ann=1.0, reg=0.0):
    feats = batch["features"]
    nc = cfg.num_captions
    B = feats.shape[0]
    feats = feats.unsqueeze(1).repeat(1, nc, 1).reshape(B * nc, -1)
    N = feats.shape[0]
    images_fv = feats @ P["imf_emb/kernel"] + P["imf_emb/bias"]
    use_ci = cfg.use_c_v or cfg.prior in ("GMM", "AG")
    ci = batch["c_v"] if use_ci else None
    ci_emb = ci @ P["cv_emb/kernel"] + P["cv_emb/bias"] if use_ci else None
    lengths = batch["lengths"]
    kld = torch.zeros(())
    if not cfg.no_encoder:
        He = P[ENC + "kernel"].shape[1] // 4
        x = P["encoder/enc_embeddings"][batch["cap_enc"]]
        z0 = torch.zeros(N, He, dtype=feats.dtype)
        c, h = cell(images_fv, z0, z0, P[ENC + "kernel"], P[ENC + "bias"])
        if cfg.use_c_v:
            c, h = cell(ci_emb, c, h, P[ENC + "kernel"], P[ENC + "bias"])
        _, (c, h) = dynamic_rnn(x, lengths, c, h, P[ENC + "kernel"], P[ENC + "bias"])
        if cfg.prior == "Normal":
            mean = h @ P["encoder/dense/kernel"] + P["encoder/dense/bias"]
            std = torch.exp(h @ P["encoder/dense_1/kernel"] + P["encoder/dense_1/bias"])
        else:
            sc = "encoder/gmm_ll_%d/" if cfg.prior == "GMM" else "encoder/ag_ll_%d/"
            tm = torch.stack([h @ P[sc % k + "dense/kernel"] + P[sc % k + "dense/bias"] for k in range(90)], 1)
            tl = torch.stack([h @ P[sc % k + "dense_1/kernel"] + P[sc % k + "dense_1/bias"] for k in range(90)], 1)
            if cfg.prior == "GMM":
                idx = noise["gmm_idx"].long()
                mean = tm[torch.arange(N), idx]
                std = torch.exp(tl)[torch.arange(N), idx]
            else:
                mean = torch.bmm(ci.unsqueeze(1), tm).squeeze(1)
                std = torch.bmm(ci.unsqueeze(1), torch.exp(tl)).squeeze(1)
        z = mean.unsqueeze(0) + std.unsqueeze(0) * noise["eps"]
        if cfg.prior in ("Normal", "GMM"):
            kld = -0.5 * torch.mean(torch.sum(1 + torch.log(std ** 2 + 0.00001) - mean ** 2 - std ** 2, 1))
        else:
            c_sigma = torch.tensor(0.1, dtype=feats.dtype)
            kc = 0.5 + torch.log(std + 0.00001) - torch.log(c_sigma + 0.00001) - (
                (mean - ci @ noise["c_means"]) ** 2 + std ** 2) / (2 * c_sigma ** 2 + 0.0000001)
            kld = -0.5 * torch.sum(kc, 1)
    Hd = P[DEC + "kernel"].shape[1] // 4
    x = P["decoder/net/dec_embeddings"][batch["cap_dec"]]
    if cfg.dec_keep_rate < 1:
        x = x * noise["drop_in"].permute(1, 0, 2) / cfg.dec_keep_rate
    z0 = torch.zeros(N, Hd, dtype=feats.dtype)
    c, h = cell(images_fv, z0, z0, P[DEC + "kernel"], P[DEC + "bias"])
    if cfg.use_c_v:
        c, h = cell(ci_emb, c, h, P[DEC + "kernel"], P[DEC + "bias"])
    if not cfg.no_encoder:
        zin = z.reshape(-1, cfg.latent_size * cfg.gen_z_samples)
        z_dec = zin @ P["decoder/net/z_rnn/kernel"] + P["decoder/net/z_rnn/bias"]
        c, h = cell(z_dec, c, h, P[DEC + "kernel"], P[DEC + "bias"])
    outs, _ = dynamic_rnn(x, lengths, c, h, P[DEC + "kernel"], P[DEC + "bias"])
    if cfg.dec_lstm_drop < 1:
        outs = outs * noise["drop_out"].permute(1, 0, 2) / cfg.dec_lstm_drop
    logits = outs.reshape(-1, Hd) @ P["decoder/rnn_logits/kernel"] + P["decoder/rnn_logits/bias"]
    labels = batch["cap_enc"].reshape(-1).long()
    ce = torch.nn.functional.cross_entropy(logits, labels, reduction="none")
    mask = torch.sign(labels.to(feats.dtype))
    rec = torch.sum(ce * mask) / torch.sum(mask) + reg
    lb = rec if cfg.no_encoder else rec + ann * kld / 10
    return kld, rec, lb
