"""`Encoder` with the constructor, attributes and `q_net()` of vae_model/encoder.py:7-110, executed
eagerly by engine.CaptionEngine.fw_encode (embedding gather, init chain, length-masked LSTM, Normal /
GMM / AG heads, reparameterised sample)."""
from .. import session


def check_images_fv(images_fv):
    if images_fv is not None and not isinstance(images_fv, session.Staged):
        raise TypeError("images_fv must be what layers.dense(features, embed_size, name='imf_emb', params=params) returned (main.py:94): the "
                        "embedding is computed inside the engine, an already embedded array cannot be used")


class Encoder(object):
    def __init__(self, images_fv, captions, lengths, params):
        """images_fv: layers.dense(..., name='imf_emb'); captions: cap_enc [N, T] int (`...<EOS>`, 0-padded); lengths [N]
        (vae_model/encoder.py:8-13).  Arrays given here are what the step runs on (session.bind); None = the caller uploads the batch
        itself through Trainer.set_batch."""
        check_images_fv(images_fv)
        self.images_fv = images_fv
        self.captions = captions
        self.lengths = lengths
        self.params = params
        self.c_i = None      # cluster vectors mapped to the embedding space (set by the caller, main.py:113)
        self.c_i_ph = None   # raw cluster vectors [N, 90]
        session.stage(params, owner='encoder', cap_enc=captions, lengths=lengths)

    def q_net(self):
        """Returns (z [S, N, L] device tensor, tm_list, tl_list); tm/tl are the [N, 90, L] stacks of
        component means / log-stds for the GMM and AG priors, [] for the Normal prior."""
        tr = session.get(self.params)
        eng = tr.cap
        if session.staged(self.params):   # arrays were given to the facades: they ARE the batch of this step
            if self.c_i_ph is not None:
                session.stage(self.params, owner='encoder', c_v=self.c_i_ph)
            feats = session.bind(self.params)
            if tr.vgg is not None and tr.vgg.wd:
                tr.vgg.reg_sumsq(eng.red.data_ptr() + 12)
            eng.fw_prepare(feats)
        z = eng.fw_encode()
        tm, tl = [], []
        if self.params.prior in ("GMM", "AG"):
            L, K = self.params.latent_size, 90
            heads = eng.buf["heads"]
            tm = heads[:, :K * L].view(-1, K, L)
            tl = heads[:, K * L:].view(-1, K, L)
        self.mean, self.std = eng.buf["mean"], eng.buf["std"]
        return z, tm, tl
