"""Caption generation on device: batched greedy decoding and beam search
(vae_model/decoder.py:145-320, ops/inference.py).

The reference decodes ONE image and ONE beam per `sess.run` (batch 1, one Python<->runtime
crossing per token per beam).  Here every round advances all images x all live beams in one
batched LSTM step + logits GEMM + softmax + top-k on the GPU, and the O(beam) bookkeeping (TopN
heaps, sentence lists, length-normalised scores) runs in `vc_beam_update`, one thread per image,
with the reference's exact semantics: stable top-`beam_size` expansion, p < 1e-12 skipped, heapq
tie order, score = logprob / len**0.7 for completed captions, `<BOS>` consumed twice
(decoder.py:230-262).  The host is not involved between decoder steps.
Per-image semantics of the z input are those of batch 1: row b of the z_rnn input is the S
samples of image b (the Q1 reshape is the identity at N = 1).
"""
import numpy as np
import torch

from . import spec
from .abi import ptr as P
from .engine import K_CL, _stream
from .utils.top_n import Beam

# vae_model/decoder.py:56 -- category ids absent from MSCOCO (obj_vectors/category_index.pickle)
UN_CLUSTERS = {0, 66, 68, 69, 71, 12, 45, 83, 26, 29, 30}


class CaptionGenerator(object):
    def __init__(self, engine):
        self.e = engine
        self.p = engine.p
        self.lib = engine.lib
        self.buf = {}
        self._ones = {}
        self._whp = None        # decoder Wh in the recurrence kernel's operand order
        self._whp_fresh = False  # re-packed at the start of every init_state (the weights may have been trained in between)

    def _b(self, name, shape, dtype=torch.float32):
        t = self.buf.get(name)
        shape = tuple(int(s) for s in shape)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype:
            t = torch.zeros(shape, dtype=dtype, device=self.e.dev)
            self.buf[name] = t
        return t

    def _dev(self, a, dtype):
        if isinstance(a, torch.Tensor):
            return a.to(self.e.dev)
        return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(self.e.dev)

    def prior_mean(self, c_v):
        """decoder.py:42-71: zeros, or for the AG prior the mean of the image's cluster means
        (empty cluster vector -> every used category id; ids beyond the 90-row matrix, quirk Q16,
        are dropped)."""
        p = self.p
        B = c_v.shape[0] if c_v is not None else 0
        if p.prior != "AG" or c_v is None:
            return None
        cm = self.e.c_means.cpu().numpy()
        out = np.zeros((B, p.latent_size), np.float32)
        for b in range(B):
            idx = np.nonzero(c_v[b] > 0)[0]
            if idx.size == 0:
                idx = np.array([i for i in range(p.num_clusters + 1) if i not in UN_CLUSTERS and i < cm.shape[0]])
            out[b] = cm[idx].mean(axis=0)
        return out

    # ------------------------------------------------------------------ init chain
    def init_state(self, features, c_v=None, eps=None):
        """State after image -> (c_v) -> z (decoder.py:96-114), batched over B images.
        eps: [S, B, L] N(0,1) draws (generated on device when None)."""
        e, p, lib, st, S = self.e, self.p, self.lib, _stream(), self.e.store
        self._whp_fresh = False  # a new generation call: re-pack Wh on the first step (5 us)
        feats = self._dev(features, np.float32)
        B = feats.shape[0]
        E, Hd, L, Sm, F = p.embed_size, p.decoder_hidden, p.latent_size, p.gen_z_samples, p.cnn_feature_size
        n_init = e.n_init_d
        X = self._b("X", (n_init, B, E))
        e.gemm(0, 0, B, E, F, feats, F, S.param("imf_emb/kernel"), E, X[0], E, S.param("imf_emb/bias"))
        if e.feed_cv:
            cv = self._dev(c_v, np.float32)
            e.gemm(0, 0, B, E, K_CL, cv, K_CL, S.param("cv_emb/kernel"), E, X[1], E, S.param("cv_emb/bias"))
        if e.enc:
            z = self._b("z", (B, Sm, L))
            if eps is None:
                epsd = self._b("eps", (B, Sm, L))
                lib.vc_philox_normal_f32(st, P(epsd), epsd.numel(), e.seed * 1000003 + 17, 5 << 32, P(e.step))
            else:
                epsd = self._dev(np.ascontiguousarray(np.transpose(np.asarray(eps, np.float32), (1, 0, 2))), np.float32)
            mean = self._b("zmean", (B * Sm, L))
            pm = self.prior_mean(np.asarray(c_v) if c_v is not None else None)
            if pm is None:
                mean.zero_()
            else:
                lib.vc_tile_rows_f32(st, P(self._dev(pm, np.float32)), B, Sm, L, P(mean))
            std = self._b("zstd", (B * Sm, L))
            lib.vc_fill_f32(st, P(std), std.numel(), float(p.std))
            lib.vc_latent_sample_f32(st, 1, B * Sm, L, P(mean), P(std), P(epsd), P(z))  # decoder.py:72-74
            e.gemm(0, 0, B, E, Sm * L, z, Sm * L, S.param("decoder/net/z_rnn/kernel"), E, X[n_init - 1], E, S.param("decoder/net/z_rnn/bias"))
        act, cs, hs = self._b("act0", (n_init, B, 4 * Hd)), self._b("cs0", (n_init + 1, B, Hd)), self._b("hs0", (n_init + 1, B, Hd))
        cs[0].zero_(); hs[0].zero_()
        lens = torch.full((B,), n_init, dtype=torch.int32, device=e.dev)
        e._need_ws(lib.vc_lstm_seq_workspace_bytes(n_init, B, E, Hd))
        lib.vc_lstm_seq_fwd_f32(st, n_init, B, E, Hd, P(X), P(S.param(spec.DEC_CELL + "kernel")), P(S.param(spec.DEC_CELL + "bias")),
                                P(lens), P(act), P(cs), P(hs), P(e.ws), e.ws_bytes, e.lstm_flags)
        return cs[n_init].clone(), hs[n_init].clone()

    # ------------------------------------------------------------------ one decoder step
    def step(self, tokens, c, h, want="probs"):
        """Feed one token per row: returns (softmax probs [M, V], c', h')."""
        e, p, lib, st, S = self.e, self.p, self.lib, _stream(), self.e.store
        M = int(tokens.shape[0])
        E, Hd, V = p.embed_size, p.decoder_hidden, e.V
        x = torch.empty((M, E), dtype=torch.float32, device=e.dev)
        lib.vc_embedding_gather_f32(st, P(S.param("decoder/net/dec_embeddings")), P(tokens), M, E, V, P(x))
        W = S.param(spec.DEC_CELL + "kernel")
        gact = torch.empty((M, 4 * Hd), dtype=torch.float32, device=e.dev)
        e.gemm(0, 0, M, 4 * Hd, E, x, E, W, 4 * Hd, gact, 4 * Hd, S.param(spec.DEC_CELL + "bias"))
        c2, h2 = torch.empty_like(c), torch.empty_like(h)
        ones = self._ones.get(M)
        if ones is None:
            ones = self._ones[M] = torch.ones((M,), dtype=torch.int32, device=e.dev)
        if lib.vc_lstm_step_packed_supported(M, Hd):  # the recurrence step kernel on Wh packed once per weight version
            if not self._whp_fresh:
                if self._whp is None or self._whp.numel() != 2 * Hd * 4 * Hd:
                    self._whp = torch.empty(2 * Hd * 4 * Hd, dtype=torch.float32, device=e.dev)
                lib.vc_lstm_pack_wh_f32(st, Hd, W.data_ptr() + E * 4 * Hd * 4, P(self._whp))
                self._whp_fresh = True
            lib.vc_lstm_step_fwd_packed_f32(st, M, Hd, 0, P(h), P(c), P(self._whp), P(gact), P(ones), P(c2), P(h2))
        else:
            lib.vc_lstm_step_fwd_f32(st, M, Hd, 0, P(h), P(c), W.data_ptr() + E * 4 * Hd * 4, P(gact), P(ones), P(c2), P(h2))
        logits = torch.empty((M, V), dtype=torch.float32, device=e.dev)
        e._timed("logits_gemm", 2.0 * M * V * Hd,
                 lambda: e.gemm(0, 0, M, V, Hd, h2, Hd, S.param("decoder/rnn_logits/kernel"), e.Vp, logits, V, S.param("decoder/rnn_logits/bias")))
        if want == "logits":
            return logits, c2, h2
        probs = torch.empty_like(logits)
        lib.vc_softmax_rows_f32(st, P(logits), M, V, V, P(probs), V)
        return probs, c2, h2

    # ------------------------------------------------------------------ greedy (online_inference)
    def _trim(self, ids, eos):
        """[steps, B] token ids -> per image the tokens up to and including its first <EOS>."""
        ids = ids.cpu().numpy()
        out = []
        for b in range(ids.shape[1]):
            col = ids[:, b].tolist()
            out.append(col[:col.index(eos) + 1] if eos in col else col)
        return out

    def greedy(self, features, c_v=None, eps=None, bos=1, eos=2, max_len=None, check_every=4):
        """decoder.py:145-201 with sample_gen='greedy' for a batch of images: returns the list
        of generated token-id lists (each ends with <EOS> unless max_len was hit).  Tokens stay on the
        device; the host only asks "has every image emitted <EOS>?" every `check_every` steps."""
        max_len = max_len or self.p.gen_max_len
        c, h = self.init_state(features, c_v, eps)
        B = c.shape[0]
        tok = torch.full((B,), bos, dtype=torch.int32, device=self.e.dev)
        ids = torch.zeros((max_len, B), dtype=torch.int32, device=self.e.dev)
        steps = 0
        for it in range(max_len):
            logits, c, h = self.step(tok, c, h, want="logits")  # argmax(softmax**(1/t)/sum) == argmax(logits)
            tok = ids[it]
            self.lib.vc_argmax_rows_f32(_stream(), P(logits), B, self.e.V, self.e.V, P(tok))
            steps = it + 1
            if check_every and steps % check_every == 0 and bool((ids[:steps] == eos).any(0).all().item()):
                break
        return self._trim(ids[:steps], eos)

    def sample(self, features, c_v=None, eps=None, bos=1, eos=2, max_len=None, uniforms=None, check_every=4):
        """decoder.py:145-201 with sample_gen='sample': tokens drawn from softmax(logits / temperature)
        (tf.multinomial).  uniforms [max_len, B] in [0,1) may be injected; otherwise Philox."""
        max_len = max_len or self.p.gen_max_len
        c, h = self.init_state(features, c_v, eps)
        B = c.shape[0]
        tok = torch.full((B,), bos, dtype=torch.int32, device=self.e.dev)
        ids = torch.zeros((max_len, B), dtype=torch.int32, device=self.e.dev)
        u = torch.empty((B,), dtype=torch.float32, device=self.e.dev)
        ud = self._dev(uniforms, np.float32) if uniforms is not None else None
        steps = 0
        for it in range(max_len):
            logits, c, h = self.step(tok, c, h, want="logits")
            if ud is not None:
                u = ud[it]
            else:
                self.lib.vc_philox_uniform_f32(_stream(), P(u), B, self.e.seed * 1000003 + 29, (16 + it) << 32, P(self.e.step))
            tok = ids[it]
            self.lib.vc_multinomial_rows_f32(_stream(), P(logits), B, self.e.V, self.e.V, float(self.p.temperature), P(u), P(tok))
            steps = it + 1
            if check_every and steps % check_every == 0 and bool((ids[:steps] == eos).any(0).all().item()):
                break
        return self._trim(ids[:steps], eos)

    # ------------------------------------------------------------------ beam search
    def beam_search(self, features, c_v=None, eps=None, bos=1, eos=2, beam_size=2, max_len=None, len_norm_f=0.7, check_every=4):
        """decoder.py:203-320 for a batch of images.  Returns per image the list of
        (sentence, score) of the kept beams in descending score order.

        Rows are [B, beam_size] throughout; every round is gather-state -> LSTM step -> logits -> softmax ->
        top-k -> vc_beam_update (the TopN bookkeeping, on device), with no host synchronisation except a
        4-byte "is any beam alive" read every `check_every` rounds."""
        lib, e, st = self.lib, self.e, _stream()
        max_len = max_len or self.p.gen_max_len
        c, h = self.init_state(features, c_v, eps)
        B, Hd, V, n = c.shape[0], self.p.decoder_hidden, e.V, int(beam_size)
        dev = e.dev
        tok = torch.full((B,), bos, dtype=torch.int32, device=dev)
        _, c, h = self.step(tok, c, h, want="logits")  # :230-236 -- probabilities discarded, state kept
        M, L = B * n, max_len + 2
        i32 = dict(dtype=torch.int32, device=dev)
        f64 = dict(dtype=torch.float64, device=dev)
        pcount, ccount = torch.ones(B, **i32), torch.zeros(B, **i32)      # partial = [Beam([bos], state b, 0.0, 0.0)]
        p_score, p_logprob, p_len = torch.zeros(M, **f64), torch.zeros(M, **f64), torch.ones(M, **i32)
        sent = [torch.full((M, L), bos, **i32), torch.zeros((M, L), **i32)]
        c_score, c_logprob, c_len, c_slot = torch.zeros(M, **f64), torch.zeros(M, **f64), torch.zeros(M, **i32), torch.zeros(M, **i32)
        c_free = torch.full((B,), (1 << (n + 1)) - 1, **i32)
        c_sent = torch.zeros((B * (n + 1), L), **i32)
        parent = torch.arange(B, **i32).repeat_interleave(n).contiguous()  # every row starts from its image's state
        tok = torch.full((M,), bos, **i32)
        tv, ti = torch.empty((M, n), dtype=torch.float32, device=dev), torch.empty((M, n), **i32)
        cg, hg = torch.empty((M, Hd), device=dev), torch.empty((M, Hd), device=dev)
        last = 0
        for it in range(max_len - 1):
            lib.vc_embedding_gather_f32(st, P(c), P(parent), M, Hd, c.shape[0], P(cg))
            lib.vc_embedding_gather_f32(st, P(h), P(parent), M, Hd, h.shape[0], P(hg))
            probs, c, h = self.step(tok, cg, hg)
            lib.vc_topk_rows_f32(st, P(probs), M, V, V, n, P(tv), P(ti))
            lib.vc_beam_update(st, B, n, L, int(eos), float(len_norm_f), P(tv), P(ti), P(pcount), P(ccount), P(p_score), P(p_logprob),
                               P(p_len), P(sent[it & 1]), P(sent[1 - (it & 1)]), P(c_score), P(c_logprob), P(c_len), P(c_slot),
                               P(c_free), P(c_sent), P(parent), P(tok))
            last = 1 - (it & 1)
            if check_every and (it + 1) % check_every == 0 and int(pcount.sum().item()) == 0:
                break
        pc, cc = pcount.cpu().numpy(), ccount.cpu().numpy()
        ps, pl = p_score.cpu().numpy().reshape(B, n), p_len.cpu().numpy().reshape(B, n)
        cs, cl, csl = c_score.cpu().numpy().reshape(B, n), c_len.cpu().numpy().reshape(B, n), c_slot.cpu().numpy().reshape(B, n)
        psent = sent[last].cpu().numpy().reshape(B, n, L)
        csent = c_sent.cpu().numpy().reshape(B, n + 1, L)
        res = []
        for b in range(B):
            if cc[b]:  # never mix complete and partial (:295-299)
                beams = [Beam(csent[b, csl[b, j], :cl[b, j]].tolist(), None, None, float(cs[b, j])) for j in range(cc[b])]
            else:
                beams = [Beam(psent[b, j, :pl[b, j]].tolist(), None, None, float(ps[b, j])) for j in range(pc[b])]
            beams.sort(reverse=True)  # TopN.extract(sort=True) on the heap array
            res.append([(bm.sentence, float(bm.score)) for bm in beams])
        return res
