"""Device-side training engine: the reference's one-`sess.run` training step
(main.py:84-183, 241-244) executed as a sequence of libvaecap C-ABI calls on HIP
streams.  torch is used for device memory, streams and torch.distributed only.

Layout decisions (MI355X-first):
  * every trainable tensor lives in ONE flat fp32 buffer per optimiser group (params,
    grads, Adam m/v): one fused optimiser launch, one global-norm reduction and ONE
    RCCL all-reduce over the gradient buffer per step;
  * sequences are time-major [T, N, .]: the image / c_v / z "init chain" steps
    (encoder.py:46-48, decoder.py:100-113) are just leading time steps of one
    contiguous LSTM input buffer;
  * the [S, N, L] sample buffer IS the [N, S*L] z_rnn input (quirk Q1) -- no copy;
  * the 180 GMM/AG head layers are stored as one [H, 2*90*L] matrix (split only when
    a reference-named checkpoint is exported).
"""
import contextlib
import os

import numpy as np
import torch

from . import abi, dp, spec
from .abi import ptr as P

K_CL = spec.NUM_CLUSTERS
TAIL = 64  # floats appended to the gradient buffer: scalars that ride in the all-reduce
T_DXSQ, T_CENUM, T_KLSUM = 0, 1, 2


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _round(n, m=64):
    return (n + m - 1) // m * m


class KernelTimer(object):
    """HIP-event pairs around selected kernel launches, recorded on the stream each launch goes to,
    inside the timed region of bench.py.  Kernels of one family may overlap (weight- and data-gradient
    convolutions run on two streams), so busy time is the UNION of their [start, end] intervals:
    roofline.achieved = algorithmic FLOPs / union time."""

    def __init__(self, all_gemms=False):
        self.recs = []
        self.base = None
        self.all_gemms = all_gemms   # also bracket every CaptionEngine.gemm call (tag "gemm"), not only the tagged launches

    def run(self, tag, flops, fn):
        st = torch.cuda.current_stream()
        if self.base is None:
            self.base = torch.cuda.Event(enable_timing=True)
            self.base.record(st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn()
        e1.record(st)
        self.recs.append((tag, flops, e0, e1))

    @staticmethod
    def _union(iv):
        tot, end = 0.0, -1.0
        for a, b in sorted(iv):
            if b <= end:
                continue
            tot += b - max(a, end)
            end = b
        return tot

    def summary(self, family=None):
        """{tag: launches, flops, seconds (sum of own durations)}; with `family` (list of tags) also
        returns the union busy time of the family under key '__union__'."""
        torch.cuda.synchronize()
        out, iv = {}, []
        for tag, fl, e0, e1 in self.recs:
            t0, t1 = self.base.elapsed_time(e0) * 1e-3, self.base.elapsed_time(e1) * 1e-3
            d = out.setdefault(tag, dict(launches=0, flops=0.0, seconds=0.0))
            d["launches"] += 1
            d["flops"] += fl
            d["seconds"] += t1 - t0
            if family and tag in family:
                iv.append((t0, t1))
        if family:
            out["__union__"] = self._union(iv)
        return out


def flat_offsets(entries):
    """{name: (offset, shape)} and the total length of a flat buffer holding `entries` back to back, each padded to 64 floats."""
    offsets, off = {}, 0
    for name, shape in entries:
        offsets[name] = (off, tuple(shape))
        off += _round(int(np.prod(shape)))
    return offsets, off


class FlatStore(object):
    """Named views over flat parameter / gradient / optimiser-slot buffers."""

    def __init__(self, entries, device, tail=0, grad_backing=None):
        self.entries = list(entries)
        self.offsets, off = flat_offsets(self.entries)
        self.n = off
        self.tail = tail
        self.device = device
        self.p = torch.zeros(off, dtype=torch.float32, device=device)
        self.g = grad_backing if grad_backing is not None else torch.zeros(off + tail, dtype=torch.float32, device=device)
        assert self.g.numel() == off + tail
        self.slots = {}

    def slot(self, name):
        if name not in self.slots:
            self.slots[name] = torch.zeros(self.n, dtype=torch.float32, device=self.device)
        return self.slots[name]

    def _view(self, buf, name):
        off, shape = self.offsets[name]
        return buf[off:off + int(np.prod(shape))].view(shape)

    def param(self, name):
        return self._view(self.p, name)

    def grad(self, name):
        return self._view(self.g, name)

    def offset(self, name):
        return self.offsets[name][0]

    def names(self):
        return [n for n, _ in self.entries]


def internal_caption_variables(p, vocab):
    """spec.caption_variables with (a) the 360 head tensors fused into two and (b) the
    embedding tables moved to the end (their dense gradients are excluded from the
    global norm, quirk Q5, so the norm runs over a prefix)."""
    ents = spec.caption_variables(p, vocab)
    fused = []
    heads_done = False
    for name, shape in ents:
        if "_ll_" in name:
            if not heads_done:
                He, L = p.encoder_hidden, p.latent_size
                fused += [("encoder/heads/kernel", (He, 2 * K_CL * L)), ("encoder/heads/bias", (2 * K_CL * L,))]
                heads_done = True
            continue
        fused.append((name, shape))
    # the logits layer is stored with its vocabulary dimension padded to a multiple of 4 (zero columns): its three GEMMs keep
    # 16-byte operand loads for vocabularies such as the reference's observed 11313 (checkpoints carry the unpadded arrays)
    Vp = _round(int(vocab), 4)
    fused = [(n, (sh[0], Vp)) if n == "decoder/rnn_logits/kernel" else ((n, (Vp,)) if n == "decoder/rnn_logits/bias" else (n, sh)) for n, sh in fused]
    emb = [e for e in fused if e[0].endswith("embeddings")]
    rest = [e for e in fused if not e[0].endswith("embeddings")]
    return rest + emb


class CaptionEngine(object):
    """Caption side of the graph: imf_emb / cv_emb, encoder q(z|x,I), KL, decoder p(x|z,I),
    masked CE, non_cnn_optimizer."""

    def __init__(self, p, vocab, device="cuda", lib=None, grad_backing=None, world=1, rank=0, group=None, seed=0,
                 force_collectives=False, q1_mode="global"):
        self.p = p
        self.collectives = world > 1 or force_collectives
        # sum-all-reduce used by the data-parallel branches; replaceable so that a single process can
        # emulate N ranks in tests (default: RCCL through torch.distributed)
        self.reduce_fn = lambda t: torch.distributed.all_reduce(t, group=self.group)
        self.gather_fn = lambda out, inp: torch.distributed.all_gather_into_tensor(out, inp, group=self.group)
        self.rscatter_fn = self._reduce_scatter
        # "global": the Q1 reshape mixes z samples over the GLOBAL batch exactly as a single-GPU run on the
        # concatenated batch would (all-gather of mean/std, reduce-scatter of their gradients, 1.5 MB each);
        # "tower": every rank mixes inside its own shard (what N towers of the reference graph compute).
        self.q1_mode = q1_mode
        self.V = int(vocab)
        self.Vp = _round(self.V, 4)
        self.lib = lib or abi.load()
        self.dev = device
        self.world, self.rank, self.group = world, rank, group
        self.use_ci = spec.uses_ci(p)
        self.feed_cv = bool(p.use_c_v) and self.use_ci
        self.enc = not p.no_encoder
        self.n_init_e = 1 + int(self.feed_cv)
        self.n_init_d = 1 + int(self.feed_cv) + int(self.enc)
        if p.encoder_hidden % 32 or p.decoder_hidden % 32:
            raise ValueError("enc_hid / dec_hid must be multiples of 32 (MFMA gate tile)")
        self.store = FlatStore(internal_caption_variables(p, self.V), device, tail=TAIL, grad_backing=grad_backing)
        names = self.store.names()
        emb = [n for n in names if n.endswith("embeddings")]
        self.n_dense = self.store.offset(emb[0]) if emb else self.store.n
        self.buf = {}
        self.pinned, self.copy_stream = {}, None
        self.capbuf, self.retired = {}, []   # capacity-sized scratch (index buffers); buffers replaced by larger ones
        self.fixed_inputs = False            # True once a hipGraph holds the input addresses (Trainer.capture)
        self._main_stream = None
        self.gmm_draw = False
        self.ws = None
        self.ws_bytes = 0
        # weight gradients, the global-norm clip and the optimiser have no consumer on the gradient chain (tf.gradients hands them
        # to apply_gradients only, ops/optimizers.py:13-16,37-47): with a second stream they run under the LSTM recurrences and,
        # when fine-tuning, under the VGG16 backward pass.  None = everything on the caller's stream.
        self.wgrad_stream = None
        self._off = False
        self.ws_off, self.ws_off_bytes = None, 0
        i32 = dict(dtype=torch.int32, device=device)
        f32 = dict(dtype=torch.float32, device=device)
        self.step = torch.zeros(1, **i32)
        self.scal = torch.zeros(8, **f32)     # vc_step_update outputs
        self.red = torch.zeros(8, **f32)      # [0] ce_num, [1] ce_den, [2] kl_sum, [3] reg_sumsq
        self.out = torch.zeros(4, **f32)      # rec_loss, kld, lower_bound, ann
        self.ns = torch.zeros(2, **f32)       # global norm, clip scale
        nb = self.lib.vc_sumsq_blocks()
        self.nb = nb
        self.part = torch.zeros(3 * nb + 4, **f32)
        self.inject = False
        self.seed = seed
        self.timer = None
        self.param_updates = 0   # optimiser steps taken by apply_gradients (its kernels write the parameters without torch noticing)
        # arithmetic of this engine's dense products and LSTM sequence calls: "f32" (the reference's tf.float32) or "bf16x3" (split-bf16
        # operands, f32 accumulate: an opt-in mode).  Carried by EVERY call (VC_GEMM_BF16X3 / VC_LSTM_BF16X3, ABI 4): engines of different
        # precision coexist in one process, and nothing another engine or caller does can change this one's arithmetic.
        self.precision = "f32"
        self.lstm_kernels = None   # None = the library's automatic choice; 0..3 = VC_LSTM_KERNELS(k) (A/B runs, tests)
        self.reg_scale = 0.0
        self.c_means = None
        if self.enc and p.prior == "AG":
            self.c_means = torch.from_numpy(init_clusters(K_CL, p.latent_size)).to(device)
        self.ann_on = int((not p.fine_tune) and (not p.restore) and p.ann_param > 1)  # main.py:163-170
        self.decay_steps = int(p.num_ex_per_epoch / (p.batch_size * world + 0.001) * p.num_epochs_per_decay)  # ops/optimizers.py:24-25, global batch

    def _reduce_scatter(self, out, inp):
        """sum-reduce-scatter of inp [world*n, L] into out [n, L] (RCCL; backends without the primitive,
        i.e. gloo in the 2-process single-GPU test, fall back to all-reduce + slice)."""
        if torch.distributed.get_backend(self.group) == "nccl":
            torch.distributed.reduce_scatter_tensor(out, inp, group=self.group)
        else:
            torch.distributed.all_reduce(inp, group=self.group)
            n = out.shape[0]
            out.copy_(inp[self.rank * n:(self.rank + 1) * n])

    # ---------------------------------------------------------------- parameters
    def load_params(self, named):
        """named: {reference variable name -> numpy array} (spec.caption_variables names)."""
        p = self.p
        need = [n for n, _ in spec.caption_variables(p, self.V)]
        missing = [n for n in need if n not in named]
        if missing:  # tf.train.Saver.restore raises NotFoundError for the same situation
            raise KeyError("checkpoint lacks %d variable(s) of this model (prior=%s, no_encoder=%s, c_v=%s), e.g. %s -- was it saved "
                           "with different options?" % (len(missing), p.prior, p.no_encoder, p.use_c_v, missing[0]))
        for name in self.store.names():
            if name == "encoder/heads/kernel":
                ks = [named[spec.head_scope(p.prior, k) + "dense/kernel"] for k in range(K_CL)]
                ls = [named[spec.head_scope(p.prior, k) + "dense_1/kernel"] for k in range(K_CL)]
                arr = np.concatenate(ks + ls, axis=1)
            elif name == "encoder/heads/bias":
                ks = [named[spec.head_scope(p.prior, k) + "dense/bias"] for k in range(K_CL)]
                ls = [named[spec.head_scope(p.prior, k) + "dense_1/bias"] for k in range(K_CL)]
                arr = np.concatenate(ks + ls, axis=0)
            else:
                arr = named[name]
            if name.startswith("decoder/rnn_logits/") and np.shape(arr)[-1] == self.V != self.Vp:  # zero-pad the vocabulary columns
                arr = np.concatenate([arr, np.zeros(np.shape(arr)[:-1] + (self.Vp - self.V,), np.float32)], axis=-1)
            dst = self.store.param(name)
            if tuple(np.shape(arr)) != tuple(dst.shape):
                raise ValueError("%s: checkpoint shape %s, model shape %s" % (name, tuple(np.shape(arr)), tuple(dst.shape)))
            dst.copy_(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)))

    def _export(self, getter):
        p = self.p
        out = {}
        L = p.latent_size
        for name in self.store.names():
            a = getter(name).detach().cpu().numpy().copy()
            if name == "encoder/heads/kernel":
                for k in range(K_CL):
                    out[spec.head_scope(p.prior, k) + "dense/kernel"] = a[:, k * L:(k + 1) * L].copy()
                    out[spec.head_scope(p.prior, k) + "dense_1/kernel"] = a[:, (K_CL + k) * L:(K_CL + k + 1) * L].copy()
            elif name == "encoder/heads/bias":
                for k in range(K_CL):
                    out[spec.head_scope(p.prior, k) + "dense/bias"] = a[k * L:(k + 1) * L].copy()
                    out[spec.head_scope(p.prior, k) + "dense_1/bias"] = a[(K_CL + k) * L:(K_CL + k + 1) * L].copy()
            elif name.startswith("decoder/rnn_logits/"):
                out[name] = np.ascontiguousarray(a[..., :self.V])
            else:
                out[name] = a
        return out

    def state_dict(self):
        torch.cuda.synchronize()
        return self._export(self.store.param)

    def grads_dict(self):
        torch.cuda.synchronize()
        return self._export(self.store.grad)

    # ---------------------------------------------------------------- buffers
    def _b(self, name, shape, dtype=torch.float32, zero=False):
        t = self.buf.get(name)
        shape = tuple(int(s) for s in shape)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype:
            t = torch.zeros(shape, dtype=dtype, device=self.dev)
            self.buf[name] = t
        elif zero:
            t.zero_()
        return t

    def _need_ws(self, nbytes):
        """Scratch for the launches of the CURRENT stream: (pointer, bytes).  Launches on the weight-gradient stream get their
        own buffer, since they run next to the ones of the caller's stream."""
        nb = max(int(nbytes), 1 << 20)
        if self._off:
            if nbytes > self.ws_off_bytes:
                self.ws_off = torch.empty(nb // 4 + 16, dtype=torch.float32, device=self.dev)
                self.ws_off_bytes = self.ws_off.numel() * 4
            return P(self.ws_off), self.ws_off_bytes
        if nbytes > self.ws_bytes:
            self.ws = torch.empty(nb // 4 + 16, dtype=torch.float32, device=self.dev)
            self.ws_bytes = self.ws.numel() * 4
        return P(self.ws), self.ws_bytes

    def enable_wgrad_stream(self, on=True):
        self.wgrad_stream = torch.cuda.Stream(self.dev) if on else None

    @contextlib.contextmanager
    def off_chain(self):
        """Launches inside run on the weight-gradient stream, ordered after everything the caller's stream holds so far; what
        they write is final for the caller only after join_off_chain().  Without a second stream: the caller's stream."""
        side = self.wgrad_stream
        if side is None or self._off:
            yield
            return
        side.wait_stream(torch.cuda.current_stream())
        self._off = True
        try:
            with torch.cuda.stream(side):
                yield
        finally:
            self._off = False

    def join_off_chain(self):
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)

    def set_precision(self, precision):
        if precision not in ("f32", "bf16x3"):
            raise ValueError("precision must be 'f32' or 'bf16x3', not %r" % (precision,))
        self.precision = precision

    @property
    def gemm_flags(self):
        """VC_GEMM_BF16X3 when this engine computes in split-bf16 (include/vaecap.h)"""
        return 4 if self.precision == "bf16x3" else 0

    @property
    def lstm_flags(self):
        """VC_LSTM_BF16X3 | VC_LSTM_KERNELS(k) of this engine's vc_lstm_seq_* calls"""
        return (0x10 if self.precision == "bf16x3" else 0) | (0 if self.lstm_kernels is None else int(self.lstm_kernels) + 1)

    def gemm(self, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=None, flags=0, tag="gemm"):
        """One dense product through vc_gemm_f32.  With a KernelTimer attached (bench.py) every call is bracketed by a HIP-event pair
        under `tag` ("logits_gemm" for the [T N, H] x [H, V] product, "gemm" for every other one: heads, projections, weight and
        data gradients) with its algorithmic FLOPs, on the stream it is launched on."""
        ws, nb = self._need_ws(self.lib.vc_gemm_workspace_bytes(M, N, K))
        run = lambda: self.lib.vc_gemm_f32(_stream(), ta, tb, M, N, K, P(A), lda, P(B), ldb, P(C), ldc, P(bias), flags | self.gemm_flags, ws, nb)
        if self.timer is not None and tag is not None and (tag != "gemm" or self.timer.all_gemms):
            self.timer.run(tag, 2.0 * M * N * K, run)
        else:
            run()

    def _timed(self, tag, flops, fn):
        if self.timer is not None:
            self.timer.run(tag, flops, fn)
        else:
            fn()

    def colsum(self, x, rows, cols, out, accumulate=0, ld=None):
        ws, nb = self._need_ws(self.lib.vc_colsum_workspace_bytes(rows, cols))
        self.lib.vc_colsum_f32(_stream(), P(x), rows, cols, ld or cols, P(out), accumulate, ws, nb)

    def dense_bwd_w(self, x, rows, fin, fout, dy, wname, bname, ld_dy=None):
        """dW = x^T.dy, db = colsum(dy) written straight into the flat gradient buffer (ld_dy: row pitch of dy, default fout)."""
        self.gemm(1, 0, fin, fout, rows, x, fin, dy, ld_dy or fout, self.store.grad(wname), fout)
        self.colsum(dy, rows, fout, self.store.grad(bname), ld=ld_dy)

    # ---------------------------------------------------------------- inputs
    def set_batch(self, batch, noise=None, extra=()):
        """Upload one batch (numpy, reference layout: cap_* are [N, T]).  Every host array of the batch (token ids, lengths,
        cluster vectors, features or images, injected noise) travels in ONE pinned staging buffer and ONE asynchronous H2D copy
        (ten separate copies cost a queue hand-off each: ~1 ms per step); the device buffers are typed views of the landing buffer.
        extra: further (name, array, torch dtype) items for the same copy (the Trainer's images).
        T (and with it every size) may change from batch to batch -- form_captions_batch pads to the longest caption of each
        batch: staging, landing and index buffers are sized to a high-water mark and only ever grow (tests/test_gpu_engine.py
        alternates T and compares with a synchronous upload)."""
        p = self.p
        nc = p.num_captions if p.mode == "training" else 1
        items = list(extra)
        if "features" in batch:
            items.append(("features", np.asarray(batch["features"], np.float32), torch.float32))
        cap_dec = np.asarray(batch["cap_dec"], np.int32)
        cap_enc = np.asarray(batch["cap_enc"], np.int32)
        self.N, self.T = cap_dec.shape
        self.B = self.N // nc
        self.nc = nc
        R = self.N * self.T
        items += [("cap_dec_t", cap_dec.T, torch.int32), ("cap_enc_t", cap_enc.T, torch.int32)]
        lens = np.asarray(batch["lengths"], np.int32)
        items += [("lens_e", lens + self.n_init_e, torch.int32), ("lens_d", lens + self.n_init_d, torch.int32)]
        if self.use_ci:
            items.append(("c_v", np.asarray(batch["c_v"], np.float32), torch.float32))
        was_injected = self.inject
        self.inject = noise is not None
        self.gmm_draw = False
        if noise is not None:
            for k in ("eps", "drop_in", "drop_out"):
                if k in noise:
                    items.append((k, np.asarray(noise[k], np.float32), torch.float32))
            if "gmm_idx" in noise:
                items.append(("gmm_idx", np.asarray(noise["gmm_idx"], np.int32), torch.int32))
        elif was_injected and not self.fixed_inputs:
            # views of an earlier injected batch: device-generated noise gets its own buffers (only then -- the device-noise
            # buffers otherwise persist from step to step: no re-allocation, no zero-fill, addresses stable under a hipGraph)
            for k in ("eps", "drop_in", "drop_out", "gmm_idx"):
                self.buf.pop(k, None)
        if noise is None and self.enc and p.prior == "GMM":
            # encoder.py:72: tf.multinomial(c_i_ph, 1) -- the cluster vector used as LOGITS (Q15): drawn on device in _noise()
            # (Philox uniforms + inverse CDF of softmax(c_v); no host loop, no read-back of the step counter)
            self.gmm_draw = True
        # inverted index of the token ids for the deterministic embedding gradient: stable counting sort ON DEVICE, on the copy stream
        # right behind the upload (i.e. under the previous step), into the index buffers of batch slot `slot`.  Vocabularies beyond the
        # single-workgroup scan of vc_embedding_grad_index (its LDS table: <= vc_embedding_index_max_vocab ids) build the same index on
        # the host (embedding_grad_index below, bit-identical by tests/test_gpu_ops.py) and ship it in the same copy.
        lib = self.lib
        nsub = int(lib.vc_embedding_index_max_subsegments(R, self.V, 32))
        host_index = self.V > int(lib.vc_embedding_index_max_vocab())
        if host_index:
            for key, ids in (("dec", cap_dec.T), ("enc", cap_enc.T)):
                o, s1, s2 = embedding_grad_index(ids, self.V, 32)
                s1p = np.full(nsub + 1, R, np.int32)   # sub-segments past the batch's real count are empty (seg1 == R)
                s1p[:s1.size] = s1
                items += [("order_" + key, o, torch.int32), ("seg1_" + key, s1p, torch.int32), ("seg2_" + key, s2, torch.int32)]
        views, slot, done = self._upload_pack(items)
        main = torch.cuda.current_stream()
        if self.fixed_inputs:
            # a captured hipGraph holds the addresses of the step's inputs: copy from the landing buffer into persistent tensors
            # (one device-to-device copy per item on the compute stream), build the index there as well
            main.wait_event(done)
            for name, v in views.items():
                dst = self.buf.get(name)
                if dst is None or self._is_landing_view(dst):
                    dst = self.buf[name] = torch.empty_like(v)
                if tuple(dst.shape) != tuple(v.shape) or dst.dtype != v.dtype:
                    raise ValueError("set_batch under a captured hipGraph: %s changed from %s %s to %s %s (shapes and dtypes are baked into the graph)"
                                     % (name, tuple(dst.shape), str(dst.dtype).replace("torch.", ""), tuple(v.shape), str(v.dtype).replace("torch.", "")))
                dst.copy_(v)
            if not host_index:
                self._build_index(R, nsub, "fixed", _stream())
            return
        self.buf.update(views)
        if host_index:
            main.wait_event(done)
            return
        with torch.cuda.stream(self.copy_stream):
            self._build_index(R, nsub, slot, _stream())
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        main.wait_event(done)

    def _is_landing_view(self, t):
        st = self.pinned.get("__pack__")
        return st is not None and any(t.untyped_storage().data_ptr() == d.untyped_storage().data_ptr() for d in st["devs"])

    def _build_index(self, R, nsub, slot, st):
        """vc_embedding_grad_index for both token tables into the index buffers of `slot` (capacity = high-water mark; the views bound
        to self.buf have this batch's exact sizes)."""
        lib = self.lib
        nb = lib.vc_embedding_index_workspace_bytes(R, self.V)
        iws = self._cap_buf("idx_ws", nb // 4 + 16, torch.int32)
        for key in ("dec", "enc"):
            o = self._cap_buf("order_%s_%s" % (key, slot), R, torch.int32)[:R]
            s1 = self._cap_buf("seg1_%s_%s" % (key, slot), nsub + 1, torch.int32)[:nsub + 1]
            s2 = self._cap_buf("seg2_%s_%s" % (key, slot), self.V + 1, torch.int32)
            lib.vc_embedding_grad_index(st, P(self.buf["cap_%s_t" % key]), R, self.V, 32, P(o), P(s1), P(s2), P(iws), iws.numel() * 4)
            self.buf["order_" + key], self.buf["seg1_" + key], self.buf["seg2_" + key] = o, s1, s2

    def _cap_buf(self, name, n, dtype):
        """Flat scratch buffer `name` with capacity >= n elements (high-water mark + 25 %; growth is rare).  A replaced buffer is
        RETIRED, not freed: queued work of either stream may still use it, and a block returned to the caching allocator could be
        handed out again before that work has run (released at the next host synchronisation, `_release_retired`)."""
        t = self.capbuf.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            cur, main = torch.cuda.current_stream(), self._main_stream or torch.cuda.current_stream()
            with torch.cuda.stream(main):
                t = torch.empty(int(n * 1.25) + 64, dtype=dtype, device=self.dev)
            if cur != main:
                cur.wait_stream(main)   # the block's previous owner finished in the compute stream's order
            if name in self.capbuf:
                self.retired.append(self.capbuf[name])
            self.capbuf[name] = t
        return t

    def _release_retired(self):
        """Drop the buffers that growth replaced, after both streams have drained."""
        if self.retired:
            torch.cuda.current_stream().synchronize()
            if self.copy_stream is not None:
                self.copy_stream.synchronize()
            self.retired = []

    def fix_inputs(self):
        """Make every input of the step a PERSISTENT tensor (called by Trainer.capture before the step is captured into a hipGraph):
        later set_batch calls copy into these tensors instead of re-binding views of the alternating landing buffers."""
        self.fixed_inputs = True
        for name, t in list(self.buf.items()):
            if self._is_landing_view(t):
                self.buf[name] = t.clone()
        if "cap_dec_t" in self.buf and self.V <= int(self.lib.vc_embedding_index_max_vocab()):
            R = self.N * self.T
            self._build_index(R, int(self.lib.vc_embedding_index_max_subsegments(R, self.V, 32)), "fixed", _stream())

    def _upload_pack(self, items):
        """[(name, host array, torch dtype)] -> ({name: device view}, landing slot, copy-done event), through one pinned staging buffer
        (a ring of three, so the host may run two steps ahead of the device) and one asynchronous copy on the copy stream.  All buffers
        have a CAPACITY (high-water mark of the packed size): a batch of another size re-uses them, so the ring position, the landing
        slot parity and the ordering events survive size changes."""
        arrs, offs, off = [], [], 0
        for name, a, dt in items:
            a = np.ascontiguousarray(a, dtype={torch.float32: np.float32, torch.uint8: np.uint8}.get(dt, np.int32))
            arrs.append(a)
            offs.append(off)
            off += (a.nbytes + 255) // 256 * 256
        total = max(off, 256)
        main = torch.cuda.current_stream()
        self._main_stream = main
        if self.copy_stream is None:
            self.copy_stream = torch.cuda.Stream()
        st = self.pinned.get("__pack__")
        if st is None:
            st = self.pinned["__pack__"] = dict(cap=0, k=0, j=0, e_prev=None, devs=[None, None], ring=[[None, None] for _ in range(3)])
        if total > st["cap"]:
            cap = (int(total * 1.25) + 4095) // 4096 * 4096
            # the old landing buffers may still be read by queued compute work and written by a queued copy: let both streams meet
            # before the new ones exist, and keep the old ones alive (self.retired) instead of returning them to the allocator
            self.copy_stream.wait_stream(main)
            main.wait_stream(self.copy_stream)
            self.retired += [d for d in st["devs"] if d is not None]
            st["devs"] = [torch.empty(cap, dtype=torch.uint8, device=self.dev) for _ in range(2)]
            for slot in st["ring"]:   # pinned buffers: torch's host allocator keeps a block until the copies that read it have finished
                slot[0] = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
            st["cap"] = cap
        k = st["k"]
        st["k"] = (k + 1) % 3
        pin, ev = st["ring"][k]
        if ev is not None:
            ev.synchronize()  # the copy that last read this staging buffer (three batches ago) has finished
        host = pin.numpy()
        for a, o in zip(arrs, offs):
            host[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
        # The copy runs on its own stream, UNDER the step that is still executing: it lands in the device buffer of the step before
        # that one (two landing buffers alternate), so it only has to wait for the work that was enqueued before the PREVIOUS
        # set_batch call; the compute stream then waits for the copy.
        e_now = torch.cuda.Event()
        e_now.record(main)
        slot = st["j"]
        dev = st["devs"][slot]
        st["j"] ^= 1
        if st["e_prev"] is not None:
            self.copy_stream.wait_event(st["e_prev"])
        with torch.cuda.stream(self.copy_stream):
            dev[:total].copy_(pin[:total], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        st["e_prev"] = e_now
        st["ring"][k][1] = ev
        views = {}
        for (name, _, dt), a, o in zip(items, arrs, offs):
            views[name] = dev[o:o + a.nbytes].view(dt).view(a.shape)
        return views, slot, ev

    def _noise(self):
        """Device-generated noise when none was injected (Philox, advanced by the step counter)."""
        p, lib, st = self.p, self.lib, _stream()
        S, N, L, T = p.gen_z_samples, self.N, p.latent_size, self.T
        if self.enc:
            eps = self._b("eps", (S, N, L))
            lib.vc_philox_normal_f32(st, P(eps), eps.numel(), self.seed * 1000003 + self.rank, 1 << 32, P(self.step))
        if p.dec_keep_rate < 1:
            m = self._b("drop_in", (T, N, p.embed_size))
            lib.vc_philox_bernoulli_f32(st, P(m), m.numel(), p.dec_keep_rate, self.seed * 1000003 + self.rank, 2 << 32, P(self.step))
        if p.dec_lstm_drop < 1:
            m = self._b("drop_out", (T, N, p.decoder_hidden))
            lib.vc_philox_bernoulli_f32(st, P(m), m.numel(), p.dec_lstm_drop, self.seed * 1000003 + self.rank, 3 << 32, P(self.step))
        if self.gmm_draw:  # encoder.py:72-75: component index ~ Categorical(softmax(c_v)) per row
            u = self._b("gmm_u", (N,))
            lib.vc_philox_uniform_f32(st, P(u), N, self.seed * 1000003 + self.rank, 4 << 32, P(self.step))
            lib.vc_multinomial_rows_f32(st, P(self.buf["c_v"]), N, K_CL, K_CL, 1.0, P(u), P(self._b("gmm_idx", (N,), torch.int32)))

    # ---------------------------------------------------------------- forward
    def forward(self, features=None, train=True):
        """Forward pass + (train) in-place d(loss)/d(logits).  `features` [B, F] device tensor
        (defaults to the uploaded precomputed features).  Stages (also callable one by one through
        the Encoder / Decoder facades): fw_prepare -> fw_encode -> fw_decode -> fw_loss."""
        self.fw_prepare(features, train)
        if self.enc:
            self.fw_encode()
        self.fw_decode()
        return self.fw_loss(train)

    def _dims(self):
        p = self.p
        return (self.N, self.T, self.B, self.nc, p.embed_size, p.encoder_hidden, p.decoder_hidden, p.latent_size,
                p.gen_z_samples, self.V, p.cnn_feature_size)

    def fw_prepare(self, features=None, train=True):
        """main.py:84-108: imf_emb (+ tile x nc), cv_emb; advances global_step / annealing when training."""
        p, lib, st, S = self.p, self.lib, _stream(), self.store
        N, T, B, nc, E, He, Hd, L, Sm, V, F = self._dims()
        feats = features if features is not None else self.buf["features"]
        self.feats = feats
        if not self.inject:
            self._noise()
        if train:  # global_step / annealing / lr schedules advance only on optimiser steps
            lib.vc_step_update(st, P(self.step), P(self.scal), p.learning_rate, p.cnn_lr, 0.8, 0.999, float(p.ann_param), self.ann_on, self.decay_steps)
        ann = self.scal[1:2]
        # imf_emb on the B image rows, then tile x nc (identical values to tiling first, main.py:84-94)
        imf = self._b("imf", (B, E))
        self.gemm(0, 0, B, E, F, feats, F, S.param("imf_emb/kernel"), E, imf, E, S.param("imf_emb/bias"))
        Te, Td = T + self.n_init_e, T + self.n_init_d
        Xd = self._b("Xd", (Td, N, E))
        if nc > 1:
            lib.vc_tile_rows_f32(st, P(imf), B, nc, E, P(Xd[0]))
        else:
            Xd[0].copy_(imf)
        if self.use_ci:
            cv = self.buf["c_v"]
            if self.feed_cv:
                self.gemm(0, 0, N, E, K_CL, cv, K_CL, S.param("cv_emb/kernel"), E, Xd[1], E, S.param("cv_emb/bias"))
        self.kl_sum = None
        return Xd[0]

    def fw_encode(self):
        """vae_model/encoder.py:24-110 + KL (main.py:118-145): returns z [S, N, L]."""
        self.fw_encode_stats()
        return self.fw_encode_sample()

    def fw_encode_stats(self):
        """First half of q_net: embedding gather, init chain, length-masked LSTM, heads -> mean, std."""
        p, lib, st, S = self.p, self.lib, _stream(), self.store
        N, T, B, nc, E, He, Hd, L, Sm, V, F = self._dims()
        Te = T + self.n_init_e
        Xd = self.buf["Xd"]
        cv = self.buf.get("c_v")
        Xe = self._b("Xe", (Te, N, E))
        Xe[:self.n_init_e].copy_(Xd[:self.n_init_e])
        # (KernelTimer "work" of the HBM-bound kernels = algorithmic BYTES, SURVEY.md section 8d: ids + gathered rows + written rows)
        self._timed("hbm_embedding_gather", T * N * (4.0 + 8.0 * E),
                    lambda: lib.vc_embedding_gather_f32(st, P(S.param("encoder/enc_embeddings")), P(self.buf["cap_enc_t"]), T * N, E, V, P(Xe[self.n_init_e])))
        act_e, cs_e, hs_e = self._b("act_e", (Te, N, 4 * He)), self._b("cs_e", (Te + 1, N, He)), self._b("hs_e", (Te + 1, N, He))
        ws, wsb = self._need_ws(lib.vc_lstm_seq_workspace_bytes(Te, N, E, He))
        # (cs_e[0] / hs_e[0] = the zero initial state: `_b` allocates zeros and nothing ever writes row 0)
        lib.vc_lstm_seq_fwd_f32(st, Te, N, E, He, P(Xe), P(S.param(spec.ENC_CELL + "kernel")), P(S.param(spec.ENC_CELL + "bias")),
                                P(self.buf["lens_e"]), P(act_e), P(cs_e), P(hs_e), ws, wsb, self.lstm_flags)
        hT = hs_e[Te]
        # [mean | std | non-PAD label count, 0, 0, 0]: one buffer, so that the data-parallel exchange of the global Q1 mix is ONE
        # all-gather (fw_encode_sample)
        msc = self._b("mean_std_cnt", (2 * N * L + 4,))
        mean, std = msc[:N * L].view(N, L), msc[N * L:2 * N * L].view(N, L)
        self.buf["mean"], self.buf["std"] = mean, std
        if p.prior == "Normal":
            logstd = self._b("logstd", (N, L))
            self.gemm(0, 0, N, L, He, hT, He, S.param("encoder/dense/kernel"), L, mean, L, S.param("encoder/dense/bias"))
            self.gemm(0, 0, N, L, He, hT, He, S.param("encoder/dense_1/kernel"), L, logstd, L, S.param("encoder/dense_1/bias"))
            lib.vc_exp_f32(st, P(logstd), N * L, P(std))
        else:
            heads = self._b("heads", (N, 2 * K_CL * L))
            self.gemm(0, 0, N, 2 * K_CL * L, He, hT, He, S.param("encoder/heads/kernel"), 2 * K_CL * L, heads, 2 * K_CL * L, S.param("encoder/heads/bias"))
            gmm = p.prior == "GMM"
            lib.vc_heads_mix_fwd_f32(st, N, K_CL, L, P(heads), None if gmm else P(cv), P(self.buf["gmm_idx"]) if gmm else None, P(mean), P(std))
        return mean, std

    def _q1_global(self):
        return self.collectives and self.q1_mode == "global" and self.world > 1

    def fw_encode_sample(self):
        """Second half of q_net: reparameterised sample (+ the data-parallel exchange of mean/std) and KL."""
        p, lib, st = self.p, self.lib, _stream()
        N, T, B, nc, E, He, Hd, L, Sm, V, F = self._dims()
        mean, std, cv = self.buf["mean"], self.buf["std"], self.buf.get("c_v")
        mu_p = None
        mode = 0
        if p.prior == "AG":
            mode = 1
            mu_p = self._b("mu_p", (N, L))
            self.gemm(0, 0, N, L, K_CL, cv, K_CL, self.c_means, L, mu_p, L)
        z = self._b("z", (Sm, N, L))
        self._den_gathered = False
        if self._q1_global():
            # ONE all-gather per step in front of the sample: every rank's [mean | std | count of non-PAD labels].  The count is a
            # function of the batch alone; gathered here, its sum (in rank order: identical on every rank) is the global CE
            # denominator of main.py:156-157 and fw_loss needs no collective of its own.
            Ng, W, M = N * self.world, self.world, 2 * N * L + 4
            msc = self.buf["mean_std_cnt"]
            lib.vc_count_nonzero_i32(st, P(self.buf["cap_enc_t"]), T * N, msc.data_ptr() + 8 * N * L)
            gath = self._b("mean_std_cnt_g", (W * M,))
            self.gather_fn(gath, msc)
            gv = gath.view(W, M)
            mg, sg = self._b("mean_g", (Ng, L)), self._b("std_g", (Ng, L))
            mg.view(W, N * L).copy_(gv[:, :N * L])
            sg.view(W, N * L).copy_(gv[:, N * L:2 * N * L])
            # global label count = the W gathered counts summed in rank order (a [W, 1] column sum with row pitch M)
            self.colsum(gath[2 * N * L:], W, 1, self.red[1:2], ld=M)
            self._den_gathered = True
            # this rank's z_rnn rows = flat range [rank*N*S, (rank+1)*N*S) of the global [S, Ng, L] tensor
            lib.vc_latent_sample_mixed_f32(st, Ng, L, self.rank * N * Sm, N * Sm, P(mg), P(sg), P(self.buf["eps"]), P(z))
        else:
            lib.vc_latent_sample_f32(st, Sm, N, L, P(mean), P(std), P(self.buf["eps"]), P(z))
        row_kl = self._b("row_kl", (N,))
        lib.vc_kl_rows_f32(st, N, L, mode, P(mean), P(std), P(mu_p), P(row_kl))
        lib.vc_reduce_sum_f32(st, P(row_kl), N, 1.0, self.red.data_ptr() + 8, 0)
        self.kl_sum = self.red.data_ptr() + 8
        return z

    def fw_decode(self):
        """vae_model/decoder.py:34-129 (training mode): returns the logits buffer [T*N, V]."""
        p, lib, st, S = self.p, self.lib, _stream(), self.store
        N, T, B, nc, E, He, Hd, L, Sm, V, F = self._dims()
        Td = T + self.n_init_d
        Xd = self.buf["Xd"]
        if self.enc:
            # decoder.py:109-111: z viewed as [N, S*L] (Q1) -> z_rnn -> the z step of the init chain
            z = self.buf["z"]
            zi = self.n_init_d - 1
            self.gemm(0, 0, N, E, Sm * L, z, Sm * L, S.param("decoder/net/z_rnn/kernel"), E, Xd[zi], E, S.param("decoder/net/z_rnn/bias"))
        xw = Xd[self.n_init_d]
        self._timed("hbm_embedding_gather", T * N * (4.0 + 8.0 * E),
                    lambda: lib.vc_embedding_gather_f32(st, P(S.param("decoder/net/dec_embeddings")), P(self.buf["cap_dec_t"]), T * N, E, V, P(xw)))
        if p.dec_keep_rate < 1:  # no train/eval switch in the reference (Q21)
            lib.vc_dropout_f32(st, P(xw), P(self.buf["drop_in"]), p.dec_keep_rate, T * N * E, P(xw))
        act_d, cs_d, hs_d = self._b("act_d", (Td, N, 4 * Hd)), self._b("cs_d", (Td + 1, N, Hd)), self._b("hs_d", (Td + 1, N, Hd))
        ws, wsb = self._need_ws(lib.vc_lstm_seq_workspace_bytes(Td, N, E, Hd))
        # (cs_d[0] / hs_d[0]: zero initial state, as above)
        lib.vc_lstm_seq_fwd_f32(st, Td, N, E, Hd, P(Xd), P(S.param(spec.DEC_CELL + "kernel")), P(S.param(spec.DEC_CELL + "bias")),
                                P(self.buf["lens_d"]), P(act_d), P(cs_d), P(hs_d), ws, wsb, self.lstm_flags)
        # outputs of the word steps.  (The reference zeroes outputs past the caption length; those rows
        # have PAD labels, so neither the loss nor any gradient can see the difference.)
        outs = hs_d[self.n_init_d + 1:]
        if p.dec_lstm_drop < 1:
            od = self._b("outs_drop", (T, N, Hd))
            lib.vc_dropout_f32(st, P(outs), P(self.buf["drop_out"]), p.dec_lstm_drop, T * N * Hd, P(od))
            outs = od
        self.outs = outs
        Vp = _round(V, 4)  # row pitch of the logits: a multiple of 4 keeps the register cross-entropy kernel for V = 11313
        logits = self._b("logits", (T * N, Vp))
        self.gemm(0, 0, T * N, Vp, Hd, outs, Hd, S.param("decoder/rnn_logits/kernel"), Vp, logits, Vp, S.param("decoder/rnn_logits/bias"), tag="logits_gemm")
        return logits[:, :V]  # (a view: the padding columns are not part of x_logits)

    def fw_loss(self, train=True):
        """main.py:152-177: masked CE (+ in-place gradient when train), loss scalars."""
        p, lib, st = self.p, self.lib, _stream()
        N, T, B, nc, E, He, Hd, L, Sm, V, F = self._dims()
        logits = self.buf["logits"]
        ann = self.scal[1:2]
        kl_sum = self.kl_sum
        labels = self.buf["cap_enc_t"]
        den = self.red[1:2]
        if not (self.collectives and getattr(self, "_den_gathered", False)):  # (global Q1 mix: the count came with the all-gather)
            lib.vc_count_nonzero_i32(st, P(labels), T * N, P(den))
            if self.collectives:
                self.reduce_fn(den)
        vector_loss = self.enc and p.prior == "AG"  # Q3
        Ng = N * self.world
        gscale = dp.scales(N, self.world, vector_loss)[0]
        row_loss = self._b("row_loss", (T * N,))
        self._timed("hbm_softmax_xent", 8.0 * T * N * V,  # logits read once, d(logits) written in place
                    lambda: lib.vc_softmax_xent_f32(st, P(logits), P(labels), T * N, V, _round(V, 4), P(den), gscale, P(row_loss), 1 if train else 0))
        lib.vc_reduce_sum_f32(st, P(row_loss), T * N, 1.0, P(self.red), 0)
        # Data-parallel training step: ce_num and kl_sum are needed for REPORTING only (the gradient's scales are the global count
        # and row number), so they ride in the tail of the gradient all-reduce (pack_tail) and the losses are finalised behind it
        # (apply_gradients) -- no collective of their own.  Without a backward pass (train=False) they are reduced here.
        self._losses_deferred = bool(self.collectives and train)
        if not self._losses_deferred:
            self._finalize_losses(kl_sum, Ng, ann, reduce=self.collectives)
        return self.out

    def _finalize_losses(self, kl_sum, Ng, ann, reduce=False):
        lib, st = self.lib, _stream()
        if reduce:  # ce_num and kl_sum summed over ranks (reporting only)
            r2 = self._b("red2", (2,), zero=True)
            r2[0:1].copy_(self.red[0:1])
            if kl_sum is not None:
                r2[1:2].copy_(self.red[2:3])
            self.reduce_fn(r2)
            self.red[0:1].copy_(r2[0:1])
            self.red[2:3].copy_(r2[1:2])
        reg = self.red.data_ptr() + 12 if self.reg_scale else None
        lib.vc_loss_finalize_f32(st, P(self.red), self.red.data_ptr() + 4, reg, float(self.reg_scale), kl_sum, 1.0 / Ng,
                                 P(ann) if self.enc else None, P(self.out))

    # ---------------------------------------------------------------- backward
    def backward(self, want_dfeatures=False):
        """Gradients of sum(lower_bound) w.r.t. every non-CNN variable, into store.g."""
        p, lib, st, S = self.p, self.lib, _stream(), self.store
        N, T, B, nc = self.N, self.T, self.B, self.nc
        E, He, Hd, L, Sm, V, F = p.embed_size, p.encoder_hidden, p.decoder_hidden, p.latent_size, p.gen_z_samples, self.V, p.cnn_feature_size
        Te, Td = T + self.n_init_e, T + self.n_init_d
        nid = self.n_init_d
        ann = self.scal[1:2]
        dlogits = self.buf["logits"]
        outs = self.outs
        Vp = _round(V, 4)
        # (off_chain: weight gradients go to the weight-gradient stream when there is one -- nothing below reads them, and nothing
        # below overwrites what they read: dlogits, the saved activations, dG and the dX rows of the init steps)
        dhs = self._b("dhs_d", (Td + 1, N, Hd))  # external gradient w.r.t. every decoder state; init steps stay 0
        douts = dhs[nid + 1:]
        self.gemm(0, 1, T * N, Hd, Vp, dlogits, Vp, S.param("decoder/rnn_logits/kernel"), Vp, douts, Hd)
        # The [Hd, V] kernel gradient of the logits layer (a chip-filling 65 GFLOP product at cfg4) is issued at the END of this pass:
        # next to the BPTT it takes every CU's LDS and the recurrence's first step kernel waits for the whole product (0.5 ms on the
        # gradient chain); issued last it runs under the VGG16 backward pass (cfg4 26.53 -> 26.28 ms; caption-only workloads: no
        # difference).  VC_LOGITS_DW=now: the former order (A/B runs).
        logits_dw_late = os.environ.get("VC_LOGITS_DW", "end") != "now"

        def side(fn):   # weight-gradient work on the weight-gradient stream
            with self.off_chain():
                fn()

        def logits_dw():
            self.dense_bwd_w(outs, T * N, Hd, Vp, dlogits, "decoder/rnn_logits/kernel", "decoder/rnn_logits/bias")  # padding columns of dlogits are 0
        if not logits_dw_late:
            side(logits_dw)
        if p.dec_lstm_drop < 1:
            lib.vc_dropout_f32(st, P(douts), P(self.buf["drop_out"]), p.dec_lstm_drop, T * N * Hd, P(douts))
        # running state gradients of both LSTMs: one buffer, one fill ([dH_d | dC_d | dC_e])
        dstate = self._b("dstate0", (N * (2 * Hd + He),), zero=True)
        dH, dC = dstate[:N * Hd].view(N, Hd), dstate[N * Hd:2 * N * Hd].view(N, Hd)
        dG, dXd = self._b("dG_d", (Td, N, 4 * Hd)), self._b("dXd", (Td, N, E))
        ws, wsb = self._need_ws(lib.vc_lstm_seq_workspace_bytes(Td, N, E, Hd))
        lib.vc_lstm_seq_bwd_data_f32(st, Td, N, E, Hd, P(S.param(spec.DEC_CELL + "kernel")), P(self.buf["lens_d"]), P(self.buf["act_d"]),
                                     P(self.buf["cs_d"]), P(dhs), P(dH), P(dC), P(dG), P(dXd), ws, wsb, self.lstm_flags)
        dxw = dXd[nid]
        if p.dec_keep_rate < 1:
            lib.vc_dropout_f32(st, P(dxw), P(self.buf["drop_in"]), p.dec_keep_rate, T * N * E, P(dxw))
        nb = self.nb
        def dec_w(dG=dG, dxw=dxw):
            ws, wsb = self._need_ws(lib.vc_lstm_seq_workspace_bytes(Td, N, E, Hd))
            lib.vc_lstm_seq_bwd_weights_f32(_stream(), Td, N, E, Hd, P(self.buf["Xd"]), P(self.buf["hs_d"]), P(dG),
                                            P(S.grad(spec.DEC_CELL + "kernel")), P(S.grad(spec.DEC_CELL + "bias")), ws, wsb, self.lstm_flags)
            self._embedding_grad("decoder/net/dec_embeddings", "dec", dxw)
            lib.vc_sumsq_partial_f32(_stream(), P(dxw), T * N * E, self.part.data_ptr() + nb * 4)  # IndexedSlices.values (Q5)
        side(dec_w)
        d_imfv = dXd[0]       # [N, E] gradient w.r.t. images_fv (decoder part)
        d_ci = dXd[1] if self.feed_cv else None
        if self.enc:
            zi = nid - 1
            dz_dec = dXd[zi]
            z = self.buf["z"]
            side(lambda: self.dense_bwd_w(z, N, Sm * L, E, dz_dec, "decoder/net/z_rnn/kernel", "decoder/net/z_rnn/bias"))
            dz = self._b("dz", (Sm, N, L))
            self.gemm(0, 1, N, Sm * L, E, dz_dec, E, S.param("decoder/net/z_rnn/kernel"), E, dz, Sm * L)
            mean, std = self.buf["mean"], self.buf["std"]
            dms = self._b("dmean_dstd", (2, N, L))
            dmean, dstd = dms[0], dms[1]
            Sb = Sm
            if self._q1_global():  # partial sums over this rank's q range for EVERY global row, then ONE reduce-scatter of [dmean | dstd]
                Ng, W = N * self.world, self.world
                pm, ps = self._b("dmean_g", (Ng, L)), self._b("dstd_g", (Ng, L))
                lib.vc_latent_sums_mixed_f32(st, Ng, L, self.rank * N * Sm, N * Sm, P(dz), P(self.buf["eps"]), P(pm), P(ps))
                pin = self._b("dmean_dstd_g", (W, 2, N * L))   # rank r's piece = [dmean rows of r | dstd rows of r]
                pin[:, 0].copy_(pm.view(W, N * L))
                pin[:, 1].copy_(ps.view(W, N * L))
                self.rscatter_fn(dms.view(-1), pin.view(-1))
                Sb = 0
            _, kl_n, kl_ag, _ = dp.scales(N, self.world, p.prior == "AG")
            hT = self.buf["hs_e"][Te]
            dhT = self._b("dH_e", (N, He))
            if p.prior == "Normal":
                lib.vc_latent_bwd_f32(st, Sb, N, L, 0, 1, P(dz), P(self.buf["eps"]), P(mean), P(std), None, P(ann), kl_n, P(dmean), P(dstd))
                side(lambda: (self.dense_bwd_w(hT, N, He, L, dmean, "encoder/dense/kernel", "encoder/dense/bias"),
                              self.dense_bwd_w(hT, N, He, L, dstd, "encoder/dense_1/kernel", "encoder/dense_1/bias")))
                self.gemm(0, 1, N, He, L, dmean, L, S.param("encoder/dense/kernel"), L, dhT, He)
                self.gemm(0, 1, N, He, L, dstd, L, S.param("encoder/dense_1/kernel"), L, dhT, He, None, 2)
            else:
                gmm = p.prior == "GMM"
                if gmm:
                    lib.vc_latent_bwd_f32(st, Sb, N, L, 0, 0, P(dz), P(self.buf["eps"]), P(mean), P(std), None, P(ann), kl_n, P(dmean), P(dstd))
                else:
                    lib.vc_latent_bwd_f32(st, Sb, N, L, 1, 0, P(dz), P(self.buf["eps"]), P(mean), P(std), P(self.buf["mu_p"]), P(ann), kl_ag, P(dmean), P(dstd))
                heads = self.buf["heads"]
                dheads = self._b("dheads", (N, 2 * K_CL * L))
                lib.vc_heads_mix_bwd_f32(st, N, K_CL, L, P(heads), None if gmm else P(self.buf["c_v"]), P(self.buf["gmm_idx"]) if gmm else None,
                                         P(dmean), P(dstd), P(dheads))
                side(lambda: self.dense_bwd_w(hT, N, He, 2 * K_CL * L, dheads, "encoder/heads/kernel", "encoder/heads/bias"))
                self.gemm(0, 1, N, He, 2 * K_CL * L, dheads, 2 * K_CL * L, S.param("encoder/heads/kernel"), 2 * K_CL * L, dhT, He)
            dC = self.buf["dstate0"][2 * N * Hd:].view(N, He)
            dG, dXe = self._b("dG_e", (Te, N, 4 * He)), self._b("dXe", (Te, N, E))
            ws, wsb = self._need_ws(lib.vc_lstm_seq_workspace_bytes(Te, N, E, He))
            lib.vc_lstm_seq_bwd_data_f32(st, Te, N, E, He, P(S.param(spec.ENC_CELL + "kernel")), P(self.buf["lens_e"]), P(self.buf["act_e"]),
                                         P(self.buf["cs_e"]), None, P(dhT), P(dC), P(dG), P(dXe), ws, wsb, self.lstm_flags)
            lib.vc_axpy_f32(st, 1.0, P(dXe[0]), N * E, P(d_imfv))
            if self.feed_cv:
                lib.vc_axpy_f32(st, 1.0, P(dXe[1]), N * E, P(d_ci))
            dxe = dXe[self.n_init_e]
            def enc_w(dG=dG, dxe=dxe):
                ws, wsb = self._need_ws(lib.vc_lstm_seq_workspace_bytes(Te, N, E, He))
                lib.vc_lstm_seq_bwd_weights_f32(_stream(), Te, N, E, He, P(self.buf["Xe"]), P(self.buf["hs_e"]), P(dG),
                                                P(S.grad(spec.ENC_CELL + "kernel")), P(S.grad(spec.ENC_CELL + "bias")), ws, wsb, self.lstm_flags)
                self._embedding_grad("encoder/enc_embeddings", "enc", dxe)
                lib.vc_sumsq_partial_f32(_stream(), P(dxe), T * N * E, self.part.data_ptr() + 2 * nb * 4)
            side(enc_w)
        else:
            with self.off_chain():
                self.part[2 * nb:3 * nb].zero_()
        if logits_dw_late:
            side(logits_dw)
        dimf = self._b("dimf", (B, E))
        if nc > 1:
            lib.vc_segment_sum_rows_f32(st, P(d_imfv), B, nc, E, P(dimf), 0)
        else:
            dimf.copy_(d_imfv)
        dfe = None
        if want_dfeatures:
            dfe = self._b("dfeatures", (B, F))
            self.gemm(0, 1, B, F, E, dimf, E, S.param("imf_emb/kernel"), E, dfe, F)
        with self.off_chain():
            self.dense_bwd_w(self.feats, B, F, E, dimf, "imf_emb/kernel", "imf_emb/bias")
            if self.use_ci:
                if self.feed_cv:
                    self.dense_bwd_w(self.buf["c_v"], N, K_CL, E, d_ci, "cv_emb/kernel", "cv_emb/bias")
                else:  # variable exists but is off the loss path (tf.gradients -> None)
                    S.grad("cv_emb/kernel").zero_()
                    S.grad("cv_emb/bias").zero_()
        return dfe

    def _embedding_grad(self, name, key, dX):
        """Dense [V, E] gradient of an embedding table from the per-position rows dX, deterministic,
        two levels: sub-segments of <= 32 positions -> partial rows -> table rows."""
        lib, st, E = self.lib, _stream(), self.p.embed_size
        seg1, seg2 = self.buf["seg1_" + key], self.buf["seg2_" + key]
        nsub = seg1.numel() - 1  # the upper bound R/32 + V: sub-segments past the batch's real count are empty (seg1 == R)
        part = self._b("embpart_" + key, (max(nsub, 1), E))
        lib.vc_embedding_grad_sorted_f32(st, P(part), P(self.buf["order_" + key]), P(seg1), E, nsub, P(dX))
        lib.vc_embedding_grad_sorted_f32(st, P(self.store.grad(name)), None, P(seg2), E, self.V, P(part))

    # ---------------------------------------------------------------- optimiser
    def pack_tail(self):
        """Scalars that must be summed over data-parallel ranks ride in the gradient buffer's tail."""
        lib, st, nb = self.lib, _stream(), self.nb
        tail = self.store.g[self.store.n:]
        lib.vc_reduce_sum_f32(st, self.part.data_ptr() + nb * 4, 2 * nb, 1.0, P(tail), 0)  # sum ||dX||^2 (both tables)
        if getattr(self, "_losses_deferred", False):  # reporting scalars of the data-parallel step: ce_num, kl_sum
            tail[1:2].copy_(self.red[0:1])
            tail[2:3].copy_(self.red[2:3])

    @property
    def param_version(self):
        """Changes whenever the parameter values may have: torch counts in-place writes through the flat buffer or a view of it
        (load_params, checkpoint restore, p.copy_), apply_gradients counts its own optimiser kernels.  Derived operands (the
        generator's packed Wh, its vocabulary projection table) are rebuilt when this differs from the version they were built at."""
        return (self.store.p._version, self.param_updates, self.store.p.data_ptr(), self.gemm_flags)

    def apply_gradients(self):
        """non_cnn_optimizer (ops/optimizers.py:3-47): global-norm clip 5.0 + Adam / SGD / Momentum."""
        p, lib, st, S, nb = self.p, self.lib, _stream(), self.store, self.nb
        self.param_updates += 1
        if getattr(self, "_losses_deferred", False):  # the tail has been summed over the ranks: finalise the reported losses
            self._losses_deferred = False
            self.red[0:1].copy_(S.g[S.n + 1:S.n + 2])
            if self.kl_sum is not None:
                self.red[2:3].copy_(S.g[S.n + 2:S.n + 3])
            self._finalize_losses(self.kl_sum, self.N * self.world, self.scal[1:2])
        lib.vc_sumsq_partial_f32(st, P(S.g), self.n_dense, P(self.part))
        self.part[nb:nb + 1].copy_(S.g[S.n:S.n + 1])
        lib.vc_clip_finalize_f32(st, P(self.part), nb + 1, float(p.lstm_clip_by_norm), P(self.ns))
        scale = self.ns.data_ptr() + 4
        if p.optimizer == "Adam":
            self._timed("hbm_adam", 28.0 * S.n,  # p, g, m, v read; p, m, v written
                        lambda: lib.vc_adam_f32(st, P(S.p), P(S.g), P(S.slot("m")), P(S.slot("v")), S.n, P(self.scal), scale, 0.8, 0.999, 1e-8, 0.0))
        elif p.optimizer == "SGD":
            lib.vc_sgd_f32(st, P(S.p), P(S.g), S.n, self.scal.data_ptr() + 8, scale, 0.0)
        else:
            a = S.slot("a")
            lr = self.scal.data_ptr() + 8
            lib.vc_momentum_f32(st, P(S.p), P(S.g), P(a), self.n_dense, lr, scale, 0.9, 0.0, None, 0)
            for name, ids in (("decoder/net/dec_embeddings", "cap_dec_t"), ("encoder/enc_embeddings", "cap_enc_t")):
                if name not in S.offsets:
                    continue
                off, shape = S.offsets[name]
                touched = self._b("touched_" + ids, (self.V,), zero=True)
                lib.vc_mark_rows_f32(st, P(touched), P(self.buf[ids]), self.T * self.N, self.V)
                n = shape[0] * shape[1]
                lib.vc_momentum_f32(st, S.p.data_ptr() + off * 4, S.g.data_ptr() + off * 4, a.data_ptr() + off * 4, n, lr, scale, 0.9, 0.0,
                                    P(touched), shape[1])

    def losses(self):
        """(kld, rec_loss, lower_bound, annealing) as Python floats -- the fetches of main.py:241-244."""
        if getattr(self, "_losses_deferred", False):
            # data-parallel training forward: the reported scalars ride in the gradient all-reduce and are final only behind
            # apply_gradients -- `out` still holds the PREVIOUS step's values.  Evaluate with forward(train=False) instead.
            raise RuntimeError("losses of a data-parallel TRAINING forward are final only after apply_gradients() "
                               "(they ride in the gradient all-reduce); use forward(train=False) / Trainer.eval_rec_loss() to evaluate")
        o = self.out.detach().cpu().numpy()
        self._release_retired()
        return float(o[1]), float(o[0]), float(o[2]), float(o[3])


def embedding_grad_index(ids, vocab, chunk=32):
    """Inverted index for the deterministic embedding gradient.  Returns
    order [R]      stable argsort of the token ids,
    seg1 [nsub+1]  boundaries (into `order`) of sub-segments of <= `chunk` positions, never
                   crossing a token boundary,
    seg2 [V+1]     for every vocabulary row the range of sub-segments that belong to it."""
    ids = np.clip(np.asarray(ids, np.int64).reshape(-1), 0, vocab - 1)
    order = np.argsort(ids, kind="stable").astype(np.int32)
    counts = np.bincount(ids, minlength=vocab)
    nsub_per = (counts + chunk - 1) // chunk
    seg2 = np.concatenate([[0], np.cumsum(nsub_per)]).astype(np.int32)
    starts = np.concatenate([[0], np.cumsum(counts)])[:-1]
    rows = np.repeat(np.arange(vocab), nsub_per)
    within = np.arange(rows.size) - np.repeat(seg2[:-1], nsub_per)
    s1 = starts[rows] + within * chunk
    seg1 = np.concatenate([s1, [ids.size]]).astype(np.int32)
    return order, seg1, seg2


def init_clusters(num_clusters=90, latent_size=150, seed=42):
    """Cluster means of the GMM / AG priors: utils/vae_utils.py:20-27 under the numpy global
    seed 42 set by Batch_Generator.__init__ (utils/batch_gen.py:65-66): 90 draws of
    2*random_sample((1, L)) - 1, each L2-normalised.  (Host-side, one-off.)"""
    rs = np.random.RandomState(seed)
    rows = []
    for _ in range(num_clusters):
        v = 2 * rs.random_sample((1, latent_size)) - 1
        rows.append(v / np.sqrt(np.sum(v ** 2)))
    return np.squeeze(np.stack(rows).astype(np.float32))
