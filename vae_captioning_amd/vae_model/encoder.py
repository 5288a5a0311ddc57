"""`Encoder` with the constructor, attributes and `q_net()` of vae_model/encoder.py:7-110, executed
eagerly by engine.CaptionEngine.fw_encode (embedding gather, init chain, length-masked LSTM, Normal /
GMM / AG heads, reparameterised sample)."""
from .. import session


class Encoder(object):
    def __init__(self, images_fv, captions, lengths, params):
        self.images_fv = images_fv
        self.captions = captions
        self.lengths = lengths
        self.params = params
        self.c_i = None      # cluster vectors mapped to the embedding space (set by the caller, main.py:113)
        self.c_i_ph = None   # raw cluster vectors [N, 90]

    def q_net(self):
        """Returns (z [S, N, L] device tensor, tm_list, tl_list); tm/tl are the [N, 90, L] stacks of
        component means / log-stds for the GMM and AG priors, [] for the Normal prior."""
        eng = session.get(self.params).cap
        z = eng.fw_encode()
        tm, tl = [], []
        if self.params.prior in ("GMM", "AG"):
            L, K = self.params.latent_size, 90
            heads = eng.buf["heads"]
            tm = heads[:, :K * L].view(-1, K, L)
            tl = heads[:, K * L:].view(-1, K, L)
        self.mean, self.std = eng.buf["mean"], eng.buf["std"]
        return z, tm, tl
