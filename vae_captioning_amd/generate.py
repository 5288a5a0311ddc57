"""Caption generation on device: batched greedy decoding and beam search
(vae_model/decoder.py:145-320, ops/inference.py).

The reference decodes ONE image and ONE beam per `sess.run` (batch 1, one Python<->runtime
crossing per token per beam).  Here every round advances all images x all live beams in one
batched LSTM step + logits GEMM + softmax + top-k on the GPU, and the O(beam) bookkeeping (TopN
heaps, sentence lists, length-normalised scores) runs in `vc_beam_update`, one wave per image,
with the reference's exact semantics: stable top-`beam_size` expansion, p < 1e-12 skipped, heapq
tie order, score = logprob / len**0.7 for completed captions, `<BOS>` consumed twice
(decoder.py:230-262).  The host is not involved between decoder steps.
Per-image semantics of the z input are those of batch 1: row b of the z_rnn input is the S
samples of image b (the Q1 reshape is the identity at N = 1).
"""
import os
import types

import numpy as np
import torch

from . import spec
from .abi import ptr as P
from .engine import K_CL, _stream

# vae_model/decoder.py:56 -- category ids absent from MSCOCO (obj_vectors/category_index.pickle)
UN_CLUSTERS = {0, 66, 68, 69, 71, 12, 45, 83, 26, 29, 30}

PHASE_TIMES = None   # diagnostics (tools/experiments/prof_beam.py): a dict here makes beam_search synchronise at its phase boundaries and add up seconds


def _phase(name, t0):
    if PHASE_TIMES is None:
        return t0
    import time
    torch.cuda.synchronize()
    t = time.perf_counter()
    PHASE_TIMES[name] = PHASE_TIMES.get(name, 0.0) + (t - t0)
    return t


def beams_from_host(ints, dbls, io, B, n, L, last):
    """The kept beams of B images from a slice's result buffers (host copies): per image the list of (sentence, score), descending.
    ints: the int32 fields at the offsets `io` (pcount / ccount [B], p_len / c_len / c_slot [B, n], sent0 / sent1 [B, n, L],
    c_sent [B, n + 1, L]); dbls: p_score [B, n] then c_score [B, n]; last: which of sent0 / sent1 the last round wrote.
    vae_model/decoder.py:295-320: the complete captions if an image has any, else its live beams -- never mixed -- sorted by
    TopN.extract(sort=True), i.e. list.sort(reverse=True) on the heap array: descending score, equal scores in array order.
    Python lists come from whole buffers (one tolist each; slicing lists is ~5x cheaper than a numpy view + tolist per beam), and of
    the two sentence stores only what the slice needs: the pool of complete captions, the live beams, or both."""
    M = B * n
    small, sc = ints[:io["sent0"]].tolist(), dbls.tolist()
    pc, cc = small[io["pcount"]:io["pcount"] + B], small[io["ccount"]:io["ccount"] + B]
    pl, cl, csl = small[io["p_len"]:io["p_len"] + M], small[io["c_len"]:io["c_len"] + M], small[io["c_slot"]:io["c_slot"] + M]
    o = io["sent%d" % last]
    pf = None if all(cc) else ints[o:o + M * L].tolist()
    cf = ints[io["c_sent"]:io["c_sent"] + B * (n + 1) * L].tolist() if any(cc) else None
    res = []
    for b in range(B):
        r0 = b * n
        if cc[b]:
            k_, s_ = cc[b], sc[M + r0:M + r0 + cc[b]]
            rows = [(b * (n + 1) + csl[r0 + j]) * L for j in range(k_)]
            ln, src = cl[r0:r0 + k_], cf
        else:
            k_, s_ = pc[b], sc[r0:r0 + pc[b]]
            rows = [(r0 + j) * L for j in range(k_)]
            ln, src = pl[r0:r0 + k_], pf
        order = sorted(range(k_), key=s_.__getitem__, reverse=True) if k_ > 1 else range(k_)
        res.append([(src[rows[j]:rows[j] + ln[j]], s_[j]) for j in order])
    return res


class CaptionGenerator(object):
    def __init__(self, engine):
        self.e = engine
        self.p = engine.p
        self.lib = engine.lib
        self.buf = {}
        self._ones = {}
        self._whp = None        # decoder Wh in the recurrence kernel's operand order
        self._whp_version = None  # engine.param_version the pack was made at (the weights may have been trained in between)
        self._graphs = {}        # captured decode rounds (hipGraphs), keyed by shapes + the addresses they bake
        self._xproj = None       # [V, 4H] input projection of every word (beam search with many rows x rounds)
        self._xproj_version = None
        self._side = []          # extra streams of a sliced beam search
        self.slices, self.slice_rows = 2, 256   # beam search: images decoded as `slices` independent slices when each has >= slice_rows rows

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

    def _load(self, dst, a):
        """host array or tensor -> the persistent device buffer dst (same shape)"""
        if not isinstance(a, torch.Tensor):
            a = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        dst.copy_(a.reshape(dst.shape), non_blocking=True)

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
        eps: [S, B, L] N(0,1) draws (generated on device when None).
        Returns (c, h) [B, H] in PERSISTENT buffers of this generator (valid until its next call).  The dozen launches (three small
        products, the sampling, two or three LSTM steps) replay as one hipGraph from the second call of a shape on: at 32-128 images
        they are ~50 us of kernels behind ~250 us of launch calls.  VC_DECODE_GRAPH=0 keeps the eager launches."""
        e, p, lib, S = self.e, self.p, self.lib, self.e.store
        B = int(features.shape[0])
        E, Hd, L, Sm, F = p.embed_size, p.decoder_hidden, p.latent_size, p.gen_z_samples, p.cnn_feature_size
        n_init = e.n_init_d
        tag = "in%d_" % B
        feats = self._b(tag + "feats", (B, F))
        self._load(feats, features)
        X = self._b(tag + "X", (n_init, B, E))
        cv = epsd = mean = std = z = None
        have_pm = False
        if e.feed_cv:
            cv = self._b(tag + "cv", (B, K_CL))
            self._load(cv, c_v)
        if e.enc:
            z, epsd = self._b(tag + "z", (B, Sm, L)), self._b(tag + "eps", (B, Sm, L))
            if eps is not None:
                self._load(epsd, np.transpose(np.asarray(eps, np.float32), (1, 0, 2)))
            mean, std = self._b(tag + "zmean", (B * Sm, L)), self._b(tag + "zstd", (B * Sm, L))
            pm = self.prior_mean(np.asarray(c_v) if c_v is not None else None)
            have_pm = pm is not None
            if have_pm:
                pmd = self._b(tag + "pm", (B, L))
                self._load(pmd, pm)
        act, cs, hs = self._b(tag + "act0", (n_init, B, 4 * Hd)), self._b(tag + "cs0", (n_init + 1, B, Hd)), self._b(tag + "hs0", (n_init + 1, B, Hd))
        lens = self.buf.get(tag + "lens%d" % n_init)
        if lens is None:
            lens = self.buf[tag + "lens%d" % n_init] = torch.full((B,), n_init, dtype=torch.int32, device=e.dev)
        e._need_ws(lib.vc_lstm_seq_workspace_bytes(n_init, B, E, Hd))
        for sh in ((B, E, F), (B, E, K_CL), (B, E, Sm * L)):
            e._need_ws(lib.vc_gemm_workspace_bytes(*sh))

        def launches(timed):
            st, tg = _stream(), ("gemm" if timed else None)
            e.gemm(0, 0, B, E, F, feats, F, S.param("imf_emb/kernel"), E, X[0], E, S.param("imf_emb/bias"), tag=tg)
            if e.feed_cv:
                e.gemm(0, 0, B, E, K_CL, cv, K_CL, S.param("cv_emb/kernel"), E, X[1], E, S.param("cv_emb/bias"), tag=tg)
            if e.enc:
                if eps is None:
                    lib.vc_philox_normal_f32(st, P(epsd), epsd.numel(), e.seed * 1000003 + 17, 5 << 32, P(e.step))
                if have_pm:
                    lib.vc_tile_rows_f32(st, P(pmd), B, Sm, L, P(mean))
                else:
                    lib.vc_fill_f32(st, P(mean), mean.numel(), 0.0)
                lib.vc_fill_f32(st, P(std), std.numel(), float(p.std))
                lib.vc_latent_sample_f32(st, 1, B * Sm, L, P(mean), P(std), P(epsd), P(z))  # decoder.py:72-74
                e.gemm(0, 0, B, E, Sm * L, z, Sm * L, S.param("decoder/net/z_rnn/kernel"), E, X[n_init - 1], E, S.param("decoder/net/z_rnn/bias"), tag=tg)
            lib.vc_fill_f32(st, P(cs[0]), B * Hd, 0.0)
            lib.vc_fill_f32(st, P(hs[0]), B * Hd, 0.0)
            lib.vc_lstm_seq_fwd_f32(st, n_init, B, E, Hd, P(X), P(S.param(spec.DEC_CELL + "kernel")), P(S.param(spec.DEC_CELL + "bias")),
                                    P(lens), P(act), P(cs), P(hs), P(e.ws), e.ws_bytes, e.lstm_flags)

        key = self._graph_key("init", B, eps is None, have_pm, float(p.std), e.lstm_flags, e.seed,
                              tensors=[feats, X, cv, epsd, mean, std, z, act, cs, hs, lens] + ([pmd] if have_pm else []))
        graph = self._graphs.get(key)
        if graph is not None:
            graph.replay()
        else:
            launches(True)
            self._capture(key, lambda: launches(False))   # (a capture executes nothing: the eager launches above are this call's)
        return cs[n_init], hs[n_init]

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
        """decoder Wh in the recurrence step kernel's operand order, once per parameter version (engine.param_version)"""
        e, p, lib = self.e, self.p, self.lib
        E, Hd = p.embed_size, p.decoder_hidden
        if self._whp_version == e.param_version or not lib.vc_lstm_step_packed_supported(M, Hd):
            return
        if self._whp is None or self._whp.numel() != 2 * Hd * 4 * Hd:
            self._whp = torch.empty(2 * Hd * 4 * Hd, dtype=torch.float32, device=e.dev)
        lib.vc_lstm_pack_wh_f32(_stream(), Hd, e.store.param(spec.DEC_CELL + "kernel").data_ptr() + E * 4 * Hd * 4, P(self._whp))
        self._whp_version = e.param_version

    def _project_vocab(self):
        """xproj [V, 4H] = dec_embeddings . Wx + b: the LSTM input projection of EVERY word, once per parameter version
        (engine.param_version: the weights may have been trained or reloaded since the last call).  A round then looks its rows up (vc_beam_gather_f32) instead of gathering embeddings and
        multiplying: one product of V rows (0.1 ms at V = 10 000) against rows x rounds of them -- callers use it when rows x rounds >= V."""
        e, p, S = self.e, self.p, self.e.store
        E, Hd, V = p.embed_size, p.decoder_hidden, e.V
        if self._xproj_version != e.param_version:
            if self._xproj is None or tuple(self._xproj.shape) != (V, 4 * Hd):
                self._xproj = torch.empty((V, 4 * Hd), dtype=torch.float32, device=e.dev)
            e.gemm(0, 0, V, 4 * Hd, E, S.param("decoder/net/dec_embeddings"), E, S.param(spec.DEC_CELL + "kernel"), 4 * Hd, self._xproj, 4 * Hd,
                   S.param(spec.DEC_CELL + "bias"))
            self._xproj_version = e.param_version
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
        if want == "state":   # (beam search's first step: decoder.py:230-236 runs it for the state only)
            return None, c2, h2
        logits = new("logits", (M, V))
        e.gemm(0, 0, M, V, Hd, h2, Hd, S.param("decoder/rnn_logits/kernel"), e.Vp, logits, V, S.param("decoder/rnn_logits/bias"),
               tag="logits_gemm" if timed else None)   # (None inside a hipGraph capture: no timer events)
        if want == "logits":
            return logits, c2, h2
        probs = torch.empty_like(logits)
        lib.vc_softmax_rows_f32(st, P(logits), M, V, V, P(probs), V)
        return probs, c2, h2

    # ------------------------------------------------------------------ captured rounds
    def _pinned(self, name, n, dtype):
        t = self.buf.get(name)
        if t is None or t.numel() != n or t.dtype != dtype:
            t = self.buf[name] = torch.zeros(int(n), dtype=dtype).pin_memory()
        return t

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
    def _beam_part(self, k, nparts, c, h, n, L, rounds, K, bos, eos, len_norm_f, xproj, fused):
        """The persistent device state of one slice of images (vae_model/decoder.py:238-247) and its round function.  A call decodes its
        images as `nparts` independent slices on `nparts` streams (beam_search): buffers are per (slice, beam width, length) and a
        captured chunk of rounds bakes their addresses."""
        lib, e = self.lib, self.e
        B, Hd, V = int(c.shape[0]), self.p.decoder_hidden, e.V
        M, dev = B * n, e.dev
        i32, f64 = torch.int32, torch.float64
        tag = "bm%d_%d_%dof%d_" % (n, L, k, nparts)
        pt = types.SimpleNamespace(B=B, M=M, k=k)
        # everything the host reads at the end lives in TWO flat buffers (int32 fields, float64 scores): two copies into pinned memory
        # bring a slice's results back, with no gathering launches in between
        sizes = [("pcount", B), ("ccount", B), ("p_len", M), ("c_len", M), ("c_slot", M), ("sent0", M * L), ("sent1", M * L), ("c_sent", B * (n + 1) * L)]
        ibuf, dbuf = self._b(tag + "ibuf", (sum(sz for _, sz in sizes),), i32), self._b(tag + "dbuf", (4 * M,), f64)
        iv, o = {}, 0
        for name, sz in sizes:
            iv[name] = (o, ibuf[o:o + sz])
            o += sz
        pcount, ccount, p_len, c_len, c_slot = (iv[k_][1] for k_ in ("pcount", "ccount", "p_len", "c_len", "c_slot"))
        sent, c_sent = [iv["sent0"][1].view(M, L), iv["sent1"][1].view(M, L)], iv["c_sent"][1].view(B * (n + 1), L)
        p_score, c_score, p_logprob, c_logprob = dbuf[0:M], dbuf[M:2 * M], dbuf[2 * M:3 * M], dbuf[3 * M:4 * M]
        c_free = self._b(tag + "c_free", (B,), i32)
        parent, tok = self._b(tag + "parent", (M,), i32), self._b(tag + "tok", (M,), i32)
        tv, ti = self._b(tag + "tv", (M, n)), self._b(tag + "ti", (M, n), i32)
        bufs = self._round_bufs(tag, M)
        # re-initialised per call: partial = [Beam([bos], state b, 0.0, 0.0)], and every row starts from its image's state (the [B, Hd]
        # state expanded to the M rows, parent = identity) -- one launch
        lib.vc_beam_init(_stream(), B, n, L, int(bos), Hd, P(c), P(h), P(bufs["c2"]), P(bufs["h2"]), P(pcount), P(ccount), P(p_score),
                         P(p_logprob), P(p_len), P(sent[0]), P(sent[1]), P(c_score), P(c_logprob), P(c_len), P(c_slot), P(c_free), P(c_sent),
                         P(parent), P(tok))
        cg, hg = self._b(tag + "cg", (M, Hd)), self._b(tag + "hg", (M, Hd))
        alive = self._b(tag + "alive", (1,))

        def one(it, timed):
            s_ = _stream()
            # every new beam continues its parent's state and feeds its last word: three row moves, one launch
            lib.vc_beam_gather_f32(s_, P(bufs["c2"]), P(bufs["h2"]), P(parent), M, Hd, P(cg), P(hg), P(xproj), P(tok), V, 4 * Hd, P(bufs["gact"]))
            if fused:   # softmax + top-k in one read of the logits (vc_softmax_topk_rows_f32: bit-identical to the two calls)
                logits, _, _ = self.step(tok, cg, hg, want="logits", bufs=bufs, timed=timed, projected=xproj is not None)
                lib.vc_softmax_topk_rows_f32(s_, P(logits), M, V, V, n, P(tv), P(ti))
            else:
                probs, _, _ = self.step(tok, cg, hg, bufs=bufs, timed=timed, projected=xproj is not None)
                lib.vc_topk_rows_f32(s_, P(probs), M, V, V, n, P(tv), P(ti))
            lib.vc_beam_update(s_, B, n, L, int(eos), float(len_norm_f), P(tv), P(ti), P(pcount), P(ccount), P(p_score), P(p_logprob),
                               P(p_len), P(sent[it & 1]), P(sent[1 - (it & 1)]), P(c_score), P(c_logprob), P(c_len), P(c_slot),
                               P(c_free), P(c_sent), P(parent), P(tok))

        def chunk_fn():
            for r in range(K):
                one(r, False)
            lib.vc_count_nonzero_i32(_stream(), P(pcount), B, P(alive))

        pt.one, pt.chunk_fn, pt.alive, pt.pcount = one, chunk_fn, alive, pcount
        pt.key = self._graph_key("beam", B, n, L, K, int(eos), float(len_norm_f),
                                 tensors=[pcount, ccount, p_score, p_logprob, p_len, sent[0], sent[1], c_score, c_logprob, c_len, c_slot, c_free, c_sent,
                                          parent, tok, tv, ti, cg, hg, alive, xproj, self._ones_for(M)] + list(bufs.values()))
        pt.graph = self._graphs.get(pt.key) if fused else None
        pt.ibuf, pt.dbuf, pt.ioff = ibuf, dbuf, {k_: v[0] for k_, v in iv.items()}
        pt.ihost, pt.dhost = self._pinned(tag + "ihost", ibuf.numel(), i32), self._pinned(tag + "dhost", 2 * M, f64)
        pt.it, pt.last, pt.done, pt.pending = 0, 0, rounds <= 0, []
        return pt

    def beam_search(self, features, c_v=None, eps=None, bos=1, eos=2, beam_size=2, max_len=None, len_norm_f=0.7, check_every=4):
        """decoder.py:203-320 for a batch of images.  Returns per image the list of
        (sentence, score) of the kept beams in descending score order.

        Rows are [B, beam_size] throughout; every round is gather-state -> LSTM step -> logits -> softmax ->
        top-k -> vc_beam_update (the TopN bookkeeping, on device), with no host synchronisation except a
        4-byte "is any beam alive" read every `check_every` rounds.

        Images are independent, and a round is a chain of one throughput-bound kernel (the logits product) and four latency-bound
        ones (state gather, LSTM step, top-k, the heap bookkeeping: together half the round's time at 640 rows, on a fraction of
        the CUs).  A batch of >= 512 rows is therefore decoded as TWO slices of images on two streams: while one slice is in its
        latency-bound kernels the other's logits product has the CUs.  VC_DECODE_SLICES=1 keeps one slice -- same beams."""
        lib, e = self.lib, self.e
        max_len = max_len or self.p.gen_max_len
        t_ph = _phase("", 0.0)
        c, h = self.init_state(features, c_v, eps)
        t_ph = _phase("init_state", t_ph)
        B, Hd, V, n = c.shape[0], self.p.decoder_hidden, e.V, int(beam_size)
        tok0 = self._b("bm_tok0", (B,), torch.int32)
        tok0.fill_(bos)
        _, c, h = self.step(tok0, c, h, want="state", bufs=self._round_bufs("bm0_", B))  # :230-236 -- probabilities discarded, state kept
        L, rounds = max_len + 2, max_len - 1
        fused = n <= 8
        # the words' input projections from a table (rows x rounds of lookups against ONE product over the vocabulary)
        xproj = self._project_vocab() if (B * n * rounds >= V and Hd % 4 == 0 and os.environ.get("VC_DECODE_XPROJ", "1") != "0") else None
        # Rounds run as hipGraph replays of K rounds each (nine launches per round otherwise): every buffer of a slice is persistent and
        # the sentence buffers alternate with the round's parity, so a chunk that starts at an even round is the same graph every time.
        # The FIRST call of a shape runs eagerly and captures the chunks at its end (a capture executes nothing); later calls replay
        # them.  VC_DECODE_GRAPH=0 keeps the eager loop -- same kernels, same beams.
        K = int(check_every) if check_every and check_every % 2 == 0 else 4
        want = int(os.environ.get("VC_DECODE_SLICES", self.slices))
        nparts = want if (want > 1 and B % want == 0 and B * n >= self.slice_rows * want and xproj is not None) else 1
        nb = B // nparts
        if nparts > 1 and lib.vc_gemm_workspace_bytes(nb * n, V, Hd) != 0:
            nparts, nb = 1, B    # (a K-split logits product writes the engine's ONE workspace: slices on two streams would share it)
        parts = [self._beam_part(k, nparts, c[k * nb:(k + 1) * nb], h[k * nb:(k + 1) * nb], n, L, rounds, K, bos, eos, len_norm_f, xproj, fused)
                 for k in range(nparts)]
        main = torch.cuda.current_stream()
        while len(self._side) < nparts - 1:
            self._side.append(torch.cuda.Stream())
        streams = [main] + self._side[:nparts - 1]
        for s in streams[1:]:
            s.wait_stream(main)
        # "is any beam alive" without idling the GPU: after every replayed chunk the 4-byte count is copied to pinned memory behind an
        # event; the host looks at the count of the PREVIOUS chunk before it launches the next (rounds of an image whose beams have all
        # ended are no-ops of vc_beam_update, so a chunk too many changes nothing)
        per = rounds // K + 2
        host_alive = self._pinned_alive(nparts * per)

        t_ph = _phase("bos step, vocabulary projection, slice set-up", t_ph)

        def advance(pt):
            if pt.graph is not None and pt.it % 2 == 0 and pt.it + K <= rounds:
                pend = pt.pending
                if check_every and len(pend) >= 2 and pend[-2][1].query() and float(host_alive[pend[-2][0]]) == 0.0:
                    pt.done = True
                    return
                pt.graph.replay()
                pt.it += K
                pt.last = 0
                if check_every:
                    slot = pt.k * per + len(pend)
                    host_alive[slot:slot + 1].copy_(pt.alive, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    pend.append((slot, ev))
            else:
                pt.one(pt.it, True)
                pt.last = 1 - (pt.it & 1)
                pt.it += 1
                if check_every and pt.it % check_every == 0:
                    lib.vc_count_nonzero_i32(_stream(), P(pt.pcount), pt.B, P(pt.alive))
                    if pt.alive.item() == 0:
                        pt.done = True
            if pt.it >= rounds:
                pt.done = True

        while not all(pt.done for pt in parts):
            for pt, s in zip(parts, streams):
                if not pt.done:
                    with torch.cuda.stream(s):
                        advance(pt)
        for s in streams[1:]:
            main.wait_stream(s)
        t_ph = _phase("rounds", t_ph)
        # results: two asynchronous copies per slice into pinned memory, one wait
        for pt in parts:
            pt.ihost.copy_(pt.ibuf, non_blocking=True)
            pt.dhost.copy_(pt.dbuf[:2 * pt.M], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        t_ph = _phase("results: copies to pinned memory", t_ph)
        res = []
        for pt in parts:
            res += beams_from_host(pt.ihost.numpy(), pt.dhost.numpy(), pt.ioff, pt.B, n, L, pt.last)
        t_ph = _phase("results to host lists", t_ph)
        if fused and rounds > K:
            for pt in parts:
                if pt.graph is None:
                    self._capture(pt.key, pt.chunk_fn)   # (the results are on the host: the capture touches no state)
        return res
