"""The oracle's TF-semantics restatements against THIRD-PARTY implementations of the same published algorithms (CPU).

The reference's numerical graph (TensorFlow 1.x + zhusuan) cannot run here, so the oracle stays **parity unpinned** (DESIGN.md
section 2) -- these checks do not change that.  What they remove is "restated from memory" for every item a second, independently
written implementation can reach: torch's LSTM cell / packed-sequence LSTM, torch.optim, torch's gradient clipping and
cross-entropy, and google.protobuf's encoder on descriptors built from the published tensor_bundle.proto.  Each test states the
mapping between the two conventions; a wrong gate order, forget bias, masking rule, epsilon placement or wire byte fails it.
"""
import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import optim as OO

torch.manual_seed(0)


def _tf_to_torch_lstm(W, b, E, H):
    """TF LSTMCell kernel [E+H, 4H], gate blocks (i, j, f, o), forget_bias 1.0 added at run time (utils/rnn_model.py:23-51)
    -> torch LSTMCell weight_ih [4H, E], weight_hh [4H, H], gate blocks (i, f, g, o), the forget bias folded into bias_ih."""
    i, j, f, o = (slice(k * H, (k + 1) * H) for k in range(4))
    order = [i, f, j, o]
    Wt = np.concatenate([W[:, s] for s in order], axis=1)
    bt = np.concatenate([b[i], b[f] + O.FORGET_BIAS, b[j], b[o]])
    return Wt[:E].T.copy(), Wt[E:].T.copy(), bt


def test_lstm_cell_step_matches_torch_lstmcell():
    rng = np.random.default_rng(0)
    N, E, H = 7, 12, 16
    W = rng.standard_normal((E + H, 4 * H)) * 0.3
    b = rng.standard_normal(4 * H) * 0.2
    x, h0, c0 = rng.standard_normal((N, E)), rng.standard_normal((N, H)) * 0.5, rng.standard_normal((N, H)) * 0.5
    cache = O.lstm_seq_fwd(x[None], np.full(N, 1), W, b, c0=c0, h0=h0)
    cell = torch.nn.LSTMCell(E, H).double()
    wih, whh, bt = _tf_to_torch_lstm(W, b, E, H)
    with torch.no_grad():
        cell.weight_ih.copy_(torch.from_numpy(wih)); cell.weight_hh.copy_(torch.from_numpy(whh))
        cell.bias_ih.copy_(torch.from_numpy(bt)); cell.bias_hh.zero_()
    h1, c1 = cell(torch.from_numpy(x), (torch.from_numpy(h0), torch.from_numpy(c0)))
    np.testing.assert_allclose(cache["hs"][1], h1.detach().numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(cache["cs"][1], c1.detach().numpy(), rtol=0, atol=1e-12)
    # a permuted gate order or a missing forget bias is far outside that tolerance
    bad = O.lstm_seq_fwd(x[None], np.full(N, 1), W, b - np.concatenate([np.zeros(2 * H), np.ones(H), np.zeros(H)]), c0=c0, h0=h0)
    assert np.abs(bad["hs"][1] - h1.detach().numpy()).max() > 1e-3


def test_dynamic_rnn_length_masking_matches_torch_packed_sequence_forward_and_backward():
    """tf.nn.dynamic_rnn(sequence_length=...) (encoder.py:49-55, decoder.py:116-121): past a row's length the state is carried and the
    output is zero.  torch.nn.LSTM on a PackedSequence implements the same published rule independently: h_n / c_n = the state at each
    row's last valid step, padded outputs zero.  Forward states and BPTT gradients (w.r.t. inputs, weights and the initial state) agree."""
    rng = np.random.default_rng(1)
    T, N, E, H = 6, 5, 8, 12
    W = rng.standard_normal((E + H, 4 * H)) * 0.3
    b = rng.standard_normal(4 * H) * 0.2
    X = rng.standard_normal((T, N, E))
    lens = np.array([6, 3, 1, 4, 2])
    cache = O.lstm_seq_fwd(X, lens, W, b)
    lstm = torch.nn.LSTM(E, H).double()
    wih, whh, bt = _tf_to_torch_lstm(W, b, E, H)
    with torch.no_grad():
        lstm.weight_ih_l0.copy_(torch.from_numpy(wih)); lstm.weight_hh_l0.copy_(torch.from_numpy(whh))
        lstm.bias_ih_l0.copy_(torch.from_numpy(bt)); lstm.bias_hh_l0.zero_()
    Xt = torch.from_numpy(X).requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(Xt, torch.from_numpy(lens), enforce_sorted=False)
    out, (hn, cn) = lstm(packed)
    out, _ = torch.nn.utils.rnn.pad_packed_sequence(out, total_length=T)
    np.testing.assert_allclose(cache["hs"][T], hn[0].detach().numpy(), rtol=0, atol=1e-12)      # final state = state at the last valid step
    np.testing.assert_allclose(cache["cs"][T], cn[0].detach().numpy(), rtol=0, atol=1e-12)
    outs = cache["hs"][1:] * cache["mask"][:, :, None]                                           # dynamic_rnn outputs: zero past the length
    np.testing.assert_allclose(outs, out.detach().numpy(), rtol=0, atol=1e-12)
    # backward: loss = sum(outputs * G) + sum(h_T * Gh)
    G, Gh = rng.standard_normal((T, N, H)), rng.standard_normal((N, H))
    ((out * torch.from_numpy(G)).sum() + (hn[0] * torch.from_numpy(Gh)).sum()).backward()
    dhs = np.zeros((T + 1, N, H))
    dhs[1:] = G * cache["mask"][:, :, None]     # the output of an inactive step is a constant zero: no gradient enters through it
    dhs[T] += Gh
    dX, dW, db, _, _ = O.lstm_seq_bwd(cache, dhs)
    np.testing.assert_allclose(dX, Xt.grad.numpy(), rtol=0, atol=1e-11)
    gih, ghh = lstm.weight_ih_l0.grad.numpy(), lstm.weight_hh_l0.grad.numpy()                   # [4H, E], [4H, H] in torch's gate order
    i, j, f, o = (slice(k * H, (k + 1) * H) for k in range(4))
    back = {0: i, 1: f, 2: j, 3: o}   # torch block k holds TF block back[k]
    for k in range(4):
        tb = slice(k * H, (k + 1) * H)
        np.testing.assert_allclose(dW[:E, back[k]], gih[tb].T, rtol=0, atol=1e-10)
        np.testing.assert_allclose(dW[E:, back[k]], ghh[tb].T, rtol=0, atol=1e-10)
        np.testing.assert_allclose(db[back[k]], lstm.bias_ih_l0.grad.numpy()[tb], rtol=0, atol=1e-10)


def test_adam_matches_torch_optim_adam_at_the_epsilon_equivalence():
    """TF: var -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps).  torch: var -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps').
    They are the same update when eps' = eps / sqrt(1-b2^t) -- so torch.optim.Adam stepped with that per-step epsilon must reproduce
    the oracle (ops/optimizers.py:37-40: beta1 = 0.8), and with a FIXED eps it must not (at gradients near eps)."""
    rng = np.random.default_rng(2)
    w0 = rng.standard_normal((6, 4)).astype(np.float64)
    grads = [rng.standard_normal(w0.shape) * 1e-7 for _ in range(5)]     # |g| ~ 1e-7: the epsilon placement matters
    lr, b1, b2, eps = 5e-4, 0.8, 0.999, 1e-8
    P, st = {"w": w0.astype(np.float32)}, {}
    wt = torch.tensor(w0.astype(np.float32), dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([wt], lr=lr, betas=(b1, b2), eps=eps)
    wfix = torch.tensor(w0.astype(np.float32), dtype=torch.float64, requires_grad=True)
    optfix = torch.optim.Adam([wfix], lr=lr, betas=(b1, b2), eps=eps)
    for t, g in enumerate(grads, 1):
        OO.adam_step(P, {"w": g.astype(np.float32)}, st, lr, t, beta1=b1, beta2=b2, eps=eps)
        for group in opt.param_groups:
            group["eps"] = eps / np.sqrt(1 - b2 ** t)
        wt.grad = torch.from_numpy(g.astype(np.float32).astype(np.float64))
        opt.step()
        wfix.grad = wt.grad.clone()
        optfix.step()
    upd = np.abs(w0.astype(np.float32) - P["w"]).max()
    assert np.abs(P["w"] - wt.detach().numpy()).max() <= 2e-3 * upd          # fp32 oracle vs fp64 torch
    assert np.abs(P["w"] - wfix.detach().numpy()).max() > 0.05 * upd          # the other epsilon placement is visibly different


def test_momentum_and_sgd_match_torch_optim_sgd():
    """MomentumOptimizer(0.9): accum = 0.9 accum + g; var -= lr accum == torch SGD(momentum=0.9, dampening=0, nesterov=False)."""
    rng = np.random.default_rng(3)
    w0 = rng.standard_normal((5, 3)).astype(np.float32)
    grads = [rng.standard_normal(w0.shape).astype(np.float32) for _ in range(4)]
    for mom in (0.0, 0.9):
        P, st = {"w": w0.copy()}, {}
        wt = torch.tensor(w0, dtype=torch.float64, requires_grad=True)
        opt = torch.optim.SGD([wt], lr=0.05, momentum=mom)
        for g in grads:
            if mom:
                OO.momentum_step(P, {"w": g}, st, 0.05, momentum=mom)
            else:
                OO.sgd_step(P, {"w": g}, 0.05)
            wt.grad = torch.from_numpy(g.astype(np.float64))
            opt.step()
        np.testing.assert_allclose(P["w"], wt.detach().numpy(), rtol=0, atol=2e-6)


def test_clip_by_global_norm_matches_torch_clip_grad_norm():
    """tf.clip_by_global_norm(5.0): g * clip / max(norm, clip) over ALL tensors jointly (ops/optimizers.py:15-16) ==
    torch.nn.utils.clip_grad_norm_(max_norm=5) (whose 1e-6 in the denominator is below fp32 resolution here)."""
    rng = np.random.default_rng(4)
    for mag in (0.1, 30.0):    # below and above the clip
        gs = {"a": (rng.standard_normal((7, 5)) * mag).astype(np.float32), "b": (rng.standard_normal(11) * mag).astype(np.float32)}
        norm = OO.global_norm(gs, {})
        scale = OO.clip_scale(norm, 5.0)
        ps = [torch.zeros(g.shape, dtype=torch.float64, requires_grad=True) for g in gs.values()]
        for p, g in zip(ps, gs.values()):
            p.grad = torch.from_numpy(g.astype(np.float64))
        total = torch.nn.utils.clip_grad_norm_(ps, max_norm=5.0)
        assert abs(float(total) - float(norm)) <= 1e-5 * float(norm)
        for p, g in zip(ps, gs.values()):
            np.testing.assert_allclose(g * scale, p.grad.numpy(), rtol=2e-6, atol=0)
        assert (scale < 1) == (mag > 1)


def test_masked_sparse_softmax_cross_entropy_matches_torch():
    """main.py:152-158: sum(ce * sign(label)) / sum(sign(label)) == F.cross_entropy(ignore_index=0, reduction='mean') -- value and
    gradient (PAD = 0 rows contribute nothing, the mean runs over the non-PAD rows only)."""
    rng = np.random.default_rng(5)
    R, V = 40, 23
    logits = rng.standard_normal((R, V)) * 3
    labels = rng.integers(0, V, R)
    labels[rng.random(R) < 0.3] = 0
    loss, cache = O.xent_masked_fwd(logits, labels)
    d = O.xent_masked_bwd(cache, 1.0)
    lt = torch.from_numpy(logits).requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lt, torch.from_numpy(labels), ignore_index=0, reduction="mean")
    ref.backward()
    assert abs(float(loss) - float(ref.detach())) < 1e-12
    np.testing.assert_allclose(d, lt.grad.numpy(), rtol=0, atol=1e-14)


# ----------------------------------------------------------------------------- TF checkpoint protobuf messages
def _bundle_messages():
    """BundleHeaderProto / BundleEntryProto message classes from descriptors built here from the PUBLISHED .proto text
    (tensorflow/core/protobuf/tensor_bundle.proto, tensor_shape.proto, versions.proto; field numbers and types as published)."""
    pb = pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "vc_tensor_bundle_test.proto", "vctest", "proto3"

    def msg(parent, name, fields):
        m = parent.message_type.add() if hasattr(parent, "message_type") else parent.nested_type.add()
        m.name = name
        for fname, num, typ, label, tname in fields:
            f = m.field.add()
            f.name, f.number, f.type, f.label = fname, num, typ, label
            if tname:
                f.type_name = tname
        return m
    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg(fd, "VersionDef", [("producer", 1, F.TYPE_INT32, OPT, ""), ("min_consumer", 2, F.TYPE_INT32, OPT, ""), ("bad_consumers", 3, F.TYPE_INT32, REP, "")])
    shape = msg(fd, "TensorShapeProto", [("dim", 2, F.TYPE_MESSAGE, REP, ".vctest.TensorShapeProto.Dim"), ("unknown_rank", 3, F.TYPE_BOOL, OPT, "")])
    msg(shape, "Dim", [("size", 1, F.TYPE_INT64, OPT, ""), ("name", 2, F.TYPE_STRING, OPT, "")])
    msg(fd, "BundleHeaderProto", [("num_shards", 1, F.TYPE_INT32, OPT, ""), ("endianness", 2, F.TYPE_INT32, OPT, ""),   # enum LITTLE = 0 / BIG = 1: varint like int32
                                   ("version", 3, F.TYPE_MESSAGE, OPT, ".vctest.VersionDef")])
    msg(fd, "BundleEntryProto", [("dtype", 1, F.TYPE_INT32, OPT, ""),                                                   # enum DataType: varint
                                  ("shape", 2, F.TYPE_MESSAGE, OPT, ".vctest.TensorShapeProto"), ("shard_id", 3, F.TYPE_INT32, OPT, ""),
                                  ("offset", 4, F.TYPE_INT64, OPT, ""), ("size", 5, F.TYPE_INT64, OPT, ""), ("crc32c", 6, F.TYPE_FIXED32, OPT, "")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:
        fac = message_factory.MessageFactory(pool)
        get = fac.GetPrototype
    return get(pool.FindMessageTypeByName("vctest.BundleHeaderProto")), get(pool.FindMessageTypeByName("vctest.BundleEntryProto"))


def test_bundle_protos_equal_google_protobuf_encoding():
    """tf_bundle.encode_header / encode_entry hand-assemble the wire bytes; google.protobuf serialises the same messages from the
    published schema: byte-identical (deterministic serialisation writes fields in field-number order, as the hand encoder does), and
    protobuf parses what the hand encoder wrote back to the same values."""
    from vae_captioning_amd import tf_bundle as tb
    Header, Entry = _bundle_messages()
    h = Header()
    h.num_shards = 1
    h.version.producer = 1
    assert tb.encode_header(1) == h.SerializeToString(deterministic=True)
    back = Header()
    back.ParseFromString(tb.encode_header(1))
    assert back.num_shards == 1 and back.endianness == 0 and back.version.producer == 1
    DT_FLOAT, DT_INT32 = 1, 3
    cases = [(DT_FLOAT, (3, 3, 64, 128), 0, 0, 294912, 0x9a3c11f7), (DT_FLOAT, (4096,), 0, 123456789012, 16384, 1), (DT_INT32, (), 0, 7, 4, 0xffffffff),
             (DT_FLOAT, (768, 2048), 0, 2 ** 33 + 5, 6291456, 0x80000000), (DT_FLOAT, (0, 5), 0, 0, 0, 0)]
    for dtype, shape, shard, off, size, crc in cases:
        e = Entry()
        e.dtype = dtype
        e.shape.SetInParent()
        for s in shape:
            e.shape.dim.add().size = s
        e.shard_id, e.offset, e.size, e.crc32c = shard, off, size, crc
        mine = tb.encode_entry(dtype, shape, shard, off, size, crc)
        assert mine == e.SerializeToString(deterministic=True), (shape, mine.hex(), e.SerializeToString(deterministic=True).hex())
        p = Entry()
        p.ParseFromString(mine)
        assert (p.dtype, tuple(d.size for d in p.shape.dim), p.shard_id, p.offset, p.size, p.crc32c) == (dtype, tuple(shape), shard, off, size, crc)
        d = tb.decode_entry(e.SerializeToString(deterministic=True))
        assert (d["dtype"], tuple(d["shape"]), d["offset"], d["size"], d["crc32c"]) == (dtype, tuple(shape), off, size, crc)


def test_crc32c_matches_an_independent_implementation():
    """CRC-32C (Castagnoli) of the tensor bytes rides in every BundleEntryProto; check the table-driven form against the bitwise
    definition of the polynomial 0x1EDC6F41 (reflected 0x82F63B78) and the RFC 3720 vectors."""
    from vae_captioning_amd import tf_bundle as tb

    def bitwise(data):
        crc = 0xffffffff
        for byte in data:
            crc ^= byte
            for _ in range(8):
                crc = (crc >> 1) ^ (0x82F63B78 & -(crc & 1))
        return crc ^ 0xffffffff
    rng = np.random.default_rng(7)
    for n in (0, 1, 31, 32, 1000):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert tb.crc32c(data) == bitwise(data)
    assert tb.crc32c(bytes(32)) == 0x8a9136aa and tb.crc32c(b"\xff" * 32) == 0x62a8ab43 and tb.crc32c(bytes(range(32))) == 0x46dd794e


def test_reader_on_a_checkpoint_written_by_an_independent_writer(tmp_path):
    """f3 (main.py:186-191, gen_caption.py:113-115): a V2 checkpoint (`.index` + `.data-00000-of-00001`) of the reference's variable
    names WRITTEN HERE without tf_bundle's writer -- BundleHeaderProto / BundleEntryProto serialised by google.protobuf from the
    published schema, the index as a LevelDB table assembled per table_format.md the way TensorFlow's TableBuilder does it (keys
    prefix-compressed against their predecessor, a restart point every 16 entries, several data blocks, index-block keys = the last
    key of each block, which the format allows in place of a shortest separator) -- and read back by tf_bundle.read_bundle /
    list_bundle / latest_checkpoint.  What this pins: the READER follows the published formats beyond what its own writer emits
    (shared key prefixes, multi-restart blocks).  What it does not: no file here was written by TensorFlow itself (absent from this
    image) -- "self-consistent + proto-exact + format-spec reader", nothing more."""
    import struct
    from vae_captioning_amd import spec, tf_bundle as tb
    from vae_captioning_amd.utils.parameters import Parameters
    Header, Entry = _bundle_messages()
    p = Parameters()
    p.prior, p.use_c_v = "Normal", False
    p.embed_size, p.encoder_hidden, p.decoder_hidden, p.latent_size, p.gen_z_samples, p.cnn_feature_size = 8, 32, 32, 5, 3, 16
    rng = np.random.default_rng(11)
    tensors = {n: rng.standard_normal(s).astype(np.float32) for n, s in spec.caption_variables(p, 37)}
    tensors["global_step"] = np.array(1234, np.int64)                         # a scalar, another dtype
    names = sorted(tensors, key=lambda s: s.encode())
    assert len(names) > 16                                                    # more than one restart interval
    # ---- data file + entries
    u32 = lambda v: struct.pack("<I", v)
    DT = {np.dtype(np.float32): 1, np.dtype(np.int64): 9}
    items, blob = [], b""
    h = Header()
    h.num_shards = 1
    h.version.producer = 1
    items.append((b"", h.SerializeToString(deterministic=True)))
    for n in names:
        a = tensors[n]
        e = Entry()
        e.dtype = DT[a.dtype]
        e.shape.SetInParent()
        for s in a.shape:
            e.shape.dim.add().size = s
        e.offset, e.size, e.crc32c = len(blob), a.nbytes, tb.mask_crc(tb.crc32c(a.tobytes()))
        blob += a.tobytes()
        items.append((n.encode(), e.SerializeToString(deterministic=True)))

    def varint(v):
        out = b""
        while v >= 128:
            out += bytes([v & 127 | 128])
            v >>= 7
        return out + bytes([v])

    def block(entries, interval=16):
        body, restarts, prev = b"", [], b""
        for i, (k, v) in enumerate(entries):
            shared = 0
            if i % interval == 0:
                restarts.append(len(body))
            else:
                while shared < min(len(k), len(prev)) and k[shared] == prev[shared]:
                    shared += 1
            body += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
            prev = k
        return body + b"".join(u32(r) for r in restarts) + u32(len(restarts))

    out, index_entries = b"", []

    def emit(blk):
        nonlocal out
        off = len(out)
        out += blk + b"\x00" + u32(tb.mask_crc(tb.crc32c(blk + b"\x00")))
        return varint(off) + varint(len(blk))
    per = 20                                                                  # entries per data block: two restarts in a block
    for i in range(0, len(items), per):
        chunk = items[i:i + per]
        assert any(a[0][:4] == b[0][:4] for a, b in zip(chunk[1:], chunk[2:]))   # keys that share a prefix with their predecessor
        index_entries.append((chunk[-1][0], emit(block(chunk))))
    meta = emit(block([]))
    idx = emit(block(index_entries, interval=1))
    footer = meta + idx
    out += footer + b"\x00" * (40 - len(footer)) + bytes.fromhex("57fb808b247547db")
    prefix = str(tmp_path / "model.ckpt")
    open(prefix + ".index", "wb").write(out)
    open(prefix + ".data-00000-of-00001", "wb").write(blob)
    open(tmp_path / "checkpoint", "w").write('model_checkpoint_path: "model.ckpt"\nall_model_checkpoint_paths: "model.ckpt"\n')
    # ---- the product's reader
    assert tb.latest_checkpoint(str(tmp_path)) == prefix
    header, entries = tb.list_bundle(prefix)
    assert header["num_shards"] == 1 and sorted(entries, key=lambda s: s.encode()) == names
    got = tb.read_bundle(prefix)
    for n in names:
        assert got[n].dtype == tensors[n].dtype and got[n].shape == tensors[n].shape and np.array_equal(got[n], tensors[n]), n
    assert "decoder/rnn_logits/kernel" in got and got["global_step"] == 1234
