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
import os

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
        self._graphs = {}        # captured decode rounds (hipGraphs), keyed by shapes + the addresses they bake
        self._xproj = None       # [V, 4H] input projection of every word (beam search with many rows x rounds)
        self._xproj_fresh = False

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
        self._xproj_fresh = False
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
    def _round_bufs(self, tag, M):
        """Persistent buffers of one decoder step over M rows (a captured round bakes their addresses)."""
        p, e = self.p, self.e
        E, Hd, V = p.embed_size, p.decoder_hidden, e.V
        return {"x": self._b(tag + "x", (M, E)), "gact": self._b(tag + "gact", (M, 4 * Hd)), "c2": self._b(tag + "c2", (M, Hd)),
                "h2": self._b(tag + "h2", (M, Hd)), "logits": self._b(tag + "logits", (M, V))}

    def _ones_for(self, M):
        """[M] int32 ones: the "every row is active" lengths of a single decoder step (persistent: captured chunks bake the address)"""
        t = self._ones.get(M)
        if t is None:
            t = self._ones[M] = torch.ones((M,), dtype=torch.int32, device=self.e.dev)
        return t

    def _pack_wh(self, M):
        """decoder Wh in the recurrence step kernel's operand order, once per generation call (init_state clears _whp_fresh)"""
        e, p, lib = self.e, self.p, self.lib
        E, Hd = p.embed_size, p.decoder_hidden
        if self._whp_fresh or not lib.vc_lstm_step_packed_supported(M, Hd):
            return
        if self._whp is None or self._whp.numel() != 2 * Hd * 4 * Hd:
            self._whp = torch.empty(2 * Hd * 4 * Hd, dtype=torch.float32, device=e.dev)
        lib.vc_lstm_pack_wh_f32(_stream(), Hd, e.store.param(spec.DEC_CELL + "kernel").data_ptr() + E * 4 * Hd * 4, P(self._whp))
        self._whp_fresh = True

    def _project_vocab(self):
        """xproj [V, 4H] = dec_embeddings . Wx + b: the LSTM input projection of EVERY word, once per generation call (the weights may
        have been trained since the last one).  A round then looks its rows up (vc_beam_gather_f32) instead of gathering embeddings and
        multiplying: one product of V rows (0.1 ms at V = 10 000) against rows x rounds of them -- callers use it when rows x rounds >= V."""
        e, p, S = self.e, self.p, self.e.store
        E, Hd, V = p.embed_size, p.decoder_hidden, e.V
        if not self._xproj_fresh:
            if self._xproj is None or tuple(self._xproj.shape) != (V, 4 * Hd):
                self._xproj = torch.empty((V, 4 * Hd), dtype=torch.float32, device=e.dev)
            e.gemm(0, 0, V, 4 * Hd, E, S.param("decoder/net/dec_embeddings"), E, S.param(spec.DEC_CELL + "kernel"), 4 * Hd, self._xproj, 4 * Hd,
                   S.param(spec.DEC_CELL + "bias"))
            self._xproj_fresh = True
        return self._xproj

    def step(self, tokens, c, h, want="probs", bufs=None, timed=True, projected=False):
        """Feed one token per row: returns (softmax probs [M, V], c', h').  bufs (from _round_bufs): write x / gate activations / new
        state / logits into these persistent tensors instead of fresh ones (c, h must not alias bufs["c2"] / bufs["h2"]).
        projected: bufs["gact"] already holds the tokens' input projections (vc_beam_gather_f32 from _project_vocab's table)."""
        e, p, lib, st, S = self.e, self.p, self.lib, _stream(), self.e.store
        M = int(tokens.shape[0])
        E, Hd, V = p.embed_size, p.decoder_hidden, e.V
        new = lambda k, shape: bufs[k] if bufs is not None else torch.empty(shape, dtype=torch.float32, device=e.dev)
        W = S.param(spec.DEC_CELL + "kernel")
        gact = new("gact", (M, 4 * Hd))
        if not projected:
            x = new("x", (M, E))
            lib.vc_embedding_gather_f32(st, P(S.param("decoder/net/dec_embeddings")), P(tokens), M, E, V, P(x))
            e.gemm(0, 0, M, 4 * Hd, E, x, E, W, 4 * Hd, gact, 4 * Hd, S.param(spec.DEC_CELL + "bias"), tag="gemm" if timed else None)
        c2, h2 = new("c2", (M, Hd)), new("h2", (M, Hd))
        ones = self._ones_for(M)
        if lib.vc_lstm_step_packed_supported(M, Hd):  # the recurrence step kernel on Wh packed once per weight version
            self._pack_wh(M)
            lib.vc_lstm_step_fwd_packed_f32(st, M, Hd, 0, P(h), P(c), P(self._whp), P(gact), P(ones), P(c2), P(h2))
        else:
            lib.vc_lstm_step_fwd_f32(st, M, Hd, 0, P(h), P(c), W.data_ptr() + E * 4 * Hd * 4, P(gact), P(ones), P(c2), P(h2))
        logits = new("logits", (M, V))
        e.gemm(0, 0, M, V, Hd, h2, Hd, S.param("decoder/rnn_logits/kernel"), e.Vp, logits, V, S.param("decoder/rnn_logits/bias"),
               tag="logits_gemm" if timed else None)   # (None inside a hipGraph capture: no timer events)
        if want == "logits":
            return logits, c2, h2
        probs = torch.empty_like(logits)
        lib.vc_softmax_rows_f32(st, P(logits), M, V, V, P(probs), V)
        return probs, c2, h2

    # ------------------------------------------------------------------ captured rounds
    def _pinned_alive(self, n):
        t = self.buf.get("bm_host_alive")
        if t is None or t.numel() < n:
            t = self.buf["bm_host_alive"] = torch.ones(max(n, 16), dtype=torch.float32).pin_memory()
        t.fill_(1.0)
        return t

    def _graph_key(self, kind, *shape, tensors=()):
        """A captured chunk is valid while EVERY address it baked is: the parameter store, the packed Wh, the engine's workspace and
        each persistent buffer of the round (`tensors`: a buffer re-allocated by a call of another shape gets a new address, and a
        graph that still holds the old one must never be replayed)."""
        e = self.e
        return ((kind,) + tuple(shape) + (e.store.p.data_ptr(), P(self._whp) if self._whp is not None else 0, P(e.ws) if e.ws is not None else 0, e.gemm_flags)
                + tuple(P(t) if t is not None else 0 for t in tensors))

    def _capture(self, key, fn):
        """hipGraph of fn() (launches on the current stream only, no allocations, no host reads).  Returns None when graphs are off
        (VC_DECODE_GRAPH=0: A/B runs)."""
        if os.environ.get("VC_DECODE_GRAPH", "1") == "0":
            return None
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) > 16:
                self._graphs.clear()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            self._graphs[key] = g
        return g

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
        device; the host only asks "has every image emitted <EOS>?" every `check_every` steps (4 bytes, vc_eos_track_i32).
        Rounds run as hipGraph replays of `check_every` decoder steps each (embedding gather, input projection, LSTM step, logits,
        argmax, stop-word tracking: six launches per step otherwise); VC_DECODE_GRAPH=0 keeps the eager loop -- same kernels, same ids."""
        lib, e = self.lib, self.e
        max_len = max_len or self.p.gen_max_len
        c0, h0 = self.init_state(features, c_v, eps)
        B, V = c0.shape[0], e.V
        dev = e.dev
        K = int(check_every) if check_every and check_every % 2 == 0 else 4   # steps per captured chunk (even: the state ends where it started)
        ids = torch.zeros((max_len, B), dtype=torch.int32, device=dev)
        chunk = self._b("g_chunk", (K + 1, B), torch.int32)    # row 0: the token fed to the chunk's first step; rows 1..K: its outputs
        done, pending = self._b("g_done", (B,), torch.int32), self._b("g_pending", (1,))
        A, Bb = self._round_bufs("gA_", B), self._round_bufs("gB_", B)
        chunk.zero_(); done.zero_()
        chunk[0].fill_(bos)
        Bb["c2"].copy_(c0); Bb["h2"].copy_(h0)    # state before step 0 lives in set B; step r reads set (B, A, B, ...) and writes the other

        def one(r, timed):
            src, dst = (Bb, A) if r % 2 == 0 else (A, Bb)
            logits, _, _ = self.step(chunk[r], src["c2"], src["h2"], want="logits", bufs=dst, timed=timed)  # argmax(softmax**(1/t)/sum) == argmax(logits)
            lib.vc_argmax_rows_f32(_stream(), P(logits), B, V, V, P(chunk[r + 1]))
            lib.vc_eos_track_i32(_stream(), P(chunk[r + 1]), B, int(eos), P(done), P(pending))

        # the first call of a shape runs one step eagerly (it sizes the workspace, whose address a captured chunk bakes; the chunk
        # then repeats that step on unchanged inputs); every call re-packs Wh (the weights may have been trained since the last one)
        self._pack_wh(B)
        baked = [chunk, done, pending, self._ones_for(B)] + list(A.values()) + list(Bb.values())
        key = self._graph_key("greedy", B, K, int(eos), tensors=baked)
        if key not in self._graphs:
            one(0, True)
            done.zero_()
            key = self._graph_key("greedy", B, K, int(eos), tensors=baked)   # (the first step may have sized the workspace)
        graph = self._capture(key, lambda: [one(r, False) for r in range(K)])
        steps = 0
        while steps < max_len:
            k = min(K, max_len - steps)
            if graph is not None and k == K:
                graph.replay()
            else:
                for r in range(k):
                    one(r, True)
            ids[steps:steps + k].copy_(chunk[1:k + 1])
            steps += k
            if steps < max_len:
                chunk[0].copy_(chunk[k])
                if k % 2:   # (an odd remainder can only be the last chunk; kept for completeness)
                    Bb["c2"].copy_(A["c2"]); Bb["h2"].copy_(A["h2"])
                if check_every and pending.item() == 0:
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
        tok0 = self._b("bm_tok0", (B,), torch.int32)
        tok0.fill_(bos)
        _, c, h = self.step(tok0, c, h, want="logits", bufs=self._round_bufs("bm0_", B))  # :230-236 -- probabilities discarded, state kept
        M, L = B * n, max_len + 2
        i32, f64 = torch.int32, torch.float64
        # the bookkeeping of vae_model/decoder.py:238-247 as PERSISTENT device buffers (a captured chunk of rounds bakes their addresses
        # and is replayed by later calls of the same shape), re-initialised per call: partial = [Beam([bos], state b, 0.0, 0.0)]
        tag = "bm%d_%d_" % (n, L)
        pcount, ccount = self._b(tag + "pcount", (B,), i32), self._b(tag + "ccount", (B,), i32)
        p_score, p_logprob, p_len = self._b(tag + "p_score", (M,), f64), self._b(tag + "p_logprob", (M,), f64), self._b(tag + "p_len", (M,), i32)
        sent = [self._b(tag + "sent0", (M, L), i32), self._b(tag + "sent1", (M, L), i32)]
        c_score, c_logprob = self._b(tag + "c_score", (M,), f64), self._b(tag + "c_logprob", (M,), f64)
        c_len, c_slot = self._b(tag + "c_len", (M,), i32), self._b(tag + "c_slot", (M,), i32)
        c_free, c_sent = self._b(tag + "c_free", (B,), i32), self._b(tag + "c_sent", (B * (n + 1), L), i32)
        parent, tok = self._b(tag + "parent", (M,), i32), self._b(tag + "tok", (M,), i32)
        tv, ti = self._b(tag + "tv", (M, n)), self._b(tag + "ti", (M, n), i32)
        first = self._b(tag + "first", (M,), i32)
        pcount.fill_(1); ccount.zero_(); p_score.zero_(); p_logprob.zero_(); p_len.fill_(1)
        sent[0].fill_(bos); sent[1].zero_()
        c_score.zero_(); c_logprob.zero_(); c_len.zero_(); c_slot.zero_(); c_sent.zero_()
        c_free.fill_((1 << (n + 1)) - 1)
        tok.fill_(bos)
        # every row starts from its image's state: the [B, Hd] state expanded to the M rows once, then parent = identity (what
        # parent = arange(B).repeat_interleave(n) on the B-row state selects)
        bufs = self._round_bufs("bm_", M)
        torch.div(torch.arange(M, dtype=i32, device=dev), n, rounding_mode="floor", out=first)
        lib.vc_embedding_gather_f32(st, P(c), P(first), M, Hd, B, P(bufs["c2"]))
        lib.vc_embedding_gather_f32(st, P(h), P(first), M, Hd, B, P(bufs["h2"]))
        torch.arange(M, dtype=i32, device=dev, out=parent)
        cg, hg = self._b("bm_cg", (M, Hd)), self._b("bm_hg", (M, Hd))
        alive = self._b("bm_alive", (1,))
        fused = n <= 8   # softmax + top-k in one read of the logits (vc_softmax_topk_rows_f32: bit-identical to the two calls)

        rounds = max_len - 1
        # the words' input projections from a table (rows x rounds of lookups against ONE product over the vocabulary)
        xproj = self._project_vocab() if (M * rounds >= V and Hd % 4 == 0 and os.environ.get("VC_DECODE_XPROJ", "1") != "0") else None

        def one(it, timed):
            s_ = _stream()
            # every new beam continues its parent's state and feeds its last word: three row moves, one launch
            lib.vc_beam_gather_f32(s_, P(bufs["c2"]), P(bufs["h2"]), P(parent), M, Hd, P(cg), P(hg), P(xproj), P(tok), V, 4 * Hd, P(bufs["gact"]))
            if fused:
                logits, _, _ = self.step(tok, cg, hg, want="logits", bufs=bufs, timed=timed, projected=xproj is not None)
                lib.vc_softmax_topk_rows_f32(s_, P(logits), M, V, V, n, P(tv), P(ti))
            else:
                probs, _, _ = self.step(tok, cg, hg, bufs=bufs, timed=timed, projected=xproj is not None)
                lib.vc_topk_rows_f32(s_, P(probs), M, V, V, n, P(tv), P(ti))
            lib.vc_beam_update(s_, B, n, L, int(eos), float(len_norm_f), P(tv), P(ti), P(pcount), P(ccount), P(p_score), P(p_logprob),
                               P(p_len), P(sent[it & 1]), P(sent[1 - (it & 1)]), P(c_score), P(c_logprob), P(c_len), P(c_slot),
                               P(c_free), P(c_sent), P(parent), P(tok))

        # Rounds run as hipGraph replays of K rounds each (nine launches per round otherwise): every buffer above is persistent and the
        # sentence buffers alternate with the round's parity, so a chunk that starts at an even round is the same graph every time.
        # The FIRST call of a shape runs eagerly and captures the chunk at its end (a capture executes nothing); later calls replay
        # it.  VC_DECODE_GRAPH=0 keeps the eager loop -- same kernels, same beams.
        K = int(check_every) if check_every and check_every % 2 == 0 else 4

        def chunk_fn():
            for r in range(K):
                one(r, False)
            lib.vc_count_nonzero_i32(_stream(), P(pcount), B, P(alive))
        key = self._graph_key("beam", B, n, L, K, int(eos), float(len_norm_f),
                              tensors=[pcount, ccount, p_score, p_logprob, p_len, sent[0], sent[1], c_score, c_logprob, c_len, c_slot, c_free, c_sent,
                                       parent, tok, tv, ti, cg, hg, alive, xproj, self._ones_for(M)] + list(bufs.values()))
        graph = self._graphs.get(key) if fused else None
        # "is any beam alive" without idling the GPU: after every replayed chunk the 4-byte count is copied to pinned memory behind an
        # event; the host looks at the count of the PREVIOUS chunk before it launches the next (rounds of an image whose beams have all
        # ended are no-ops of vc_beam_update, so a chunk too many changes nothing)
        host_alive = self._pinned_alive(rounds // K + 2)
        pending = []
        last, it = 0, 0
        while it < rounds:
            if graph is not None and it % 2 == 0 and it + K <= rounds:
                if check_every and len(pending) >= 2 and pending[-2][1].query() and float(host_alive[pending[-2][0]]) == 0.0:
                    break
                graph.replay()
                it += K
                last = 0
                if check_every:
                    slot = len(pending)
                    host_alive[slot:slot + 1].copy_(alive, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    pending.append((slot, ev))
            else:
                one(it, True)
                last = 1 - (it & 1)
                it += 1
                if check_every and it % check_every == 0:
                    lib.vc_count_nonzero_i32(st, P(pcount), B, P(alive))
                    if alive.item() == 0:
                        break
        # results: TWO device-to-host copies (every int32 field in one buffer, the two float64 score arrays in another)
        ints = torch.cat([pcount, ccount, p_len, c_len, c_slot, sent[last].reshape(-1), c_sent.reshape(-1)]).cpu().numpy()
        dbls = torch.cat([p_score, c_score]).cpu().numpy()
        o = 0
        def take(cnt, shape):
            nonlocal o
            a = ints[o:o + cnt].reshape(shape)
            o += cnt
            return a
        pc, cc = take(B, (B,)).tolist(), take(B, (B,)).tolist()
        pl, cl, csl = take(M, (B, n)).tolist(), take(M, (B, n)).tolist(), take(M, (B, n)).tolist()
        psent, csent = take(M * L, (B, n, L)), take(B * (n + 1) * L, (B, n + 1, L))
        ps, cs = dbls[:M].reshape(B, n).tolist(), dbls[M:].reshape(B, n).tolist()
        res = []
        for b in range(B):
            if cc[b]:  # never mix complete and partial (:295-299)
                beams = [Beam(csent[b, csl[b][j], :cl[b][j]].tolist(), None, None, cs[b][j]) for j in range(cc[b])]
            else:
                beams = [Beam(psent[b, j, :pl[b][j]].tolist(), None, None, ps[b][j]) for j in range(pc[b])]
            beams.sort(reverse=True)  # TopN.extract(sort=True) on the heap array
            res.append([(bm.sentence, float(bm.score)) for bm in beams])
        if graph is None and fused and rounds > K:
            self._capture(key, chunk_fn)   # (the results are on the host: the capture touches no state)
        return res
