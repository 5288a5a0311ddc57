"""TensorFlow V2 checkpoint ("tensor bundle") reader and writer, host side.

The reference saves and restores its variables with ``tf.train.Saver`` (main.py:186-191,
286-288; gen_caption.py:113-115; ops/inference.py restores the same files), which writes

    <prefix>.index                   an SSTable: key "" -> BundleHeaderProto,
                                     key <variable name> -> BundleEntryProto
    <prefix>.data-00000-of-00001     the tensors' raw little-endian bytes, back to back in key order
    checkpoint                       text CheckpointState naming the latest prefix

TensorFlow is not installable in this image, so this module restates the published on-disk format
(tensorflow/core/util/tensor_bundle + tensorflow/core/lib/io/table, which is LevelDB's table format
with CRC-32C block trailers) and is UNVERIFIED against a TensorFlow-written file: the tests pin the
pieces that have public known answers (CRC-32C vectors and its mask, varint coding, the footer
magic, protobuf wire bytes of the two messages) and a write -> read round trip.

Layout written here == what BundleWriter produces for one shard: no block compression, 16-key
restart interval in data blocks, 1 in the index block, 256 KiB blocks, empty metaindex block,
48-byte footer ending in the magic 0xdb4775248b80fb57.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
BLOCK_SIZE = 262144
RESTART_INTERVAL = 16
# tensorflow/core/framework/types.proto
DT = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("uint8"): 4, np.dtype("int16"): 5,
      np.dtype("int8"): 6, np.dtype("int64"): 9, np.dtype("bool"): 10}
DT_INV = {v: k for k, v in DT.items()}

_CRC_TABLE = None


def _crc32c_py(data, crc=0):
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    t = _CRC_TABLE
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = (c >> 8) ^ t[(c ^ b) & 0xFF]
    return c ^ 0xFFFFFFFF


def crc32c(data, crc=0):
    """CRC-32C of bytes / a C-contiguous numpy array.  Large inputs go through libvaecap's
    vc_host_crc32c (a host function, no GPU involved); small ones through a table loop."""
    if isinstance(data, np.ndarray):
        buf = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    else:
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
    if buf.size < 4096:
        return _crc32c_py(buf.tobytes(), crc)
    import ctypes
    from . import abi
    lib = abi.load()
    out = ctypes.c_uint32(crc)
    lib.vc_host_crc32c(buf.ctypes.data, buf.size, ctypes.addressof(out))
    return out.value


def mask_crc(crc):
    """crc32c::Mask: rotate right by 15 and add a constant (so a CRC of data that embeds CRCs stays sound)."""
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def unmask_crc(m):
    rot = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


def varint(n):
    out = bytearray()
    n &= 0xFFFFFFFFFFFFFFFF
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def read_varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if b < 0x80:
            return val, pos
        shift += 7


# ----------------------------------------------------------------------------- protobuf (two messages)
def encode_header(num_shards=1):
    """BundleHeaderProto{num_shards=1; endianness=LITTLE(0, default: omitted); version{producer=1}}."""
    return b"\x08" + varint(num_shards) + b"\x1a\x02\x08\x01"


def encode_entry(dtype, shape, shard_id, offset, size, crc_masked):
    """BundleEntryProto: dtype(1) shape(2) shard_id(3) offset(4) size(5) crc32c(6, fixed32); proto3 omits zeros."""
    dims = b"".join(b"\x12" + varint(len(d)) + d for d in ((b"\x08" + varint(int(s))) if int(s) != 0 else b"" for s in shape))
    out = b"\x08" + varint(dtype) + b"\x12" + varint(len(dims)) + dims
    if shard_id:
        out += b"\x18" + varint(shard_id)
    if offset:
        out += b"\x20" + varint(offset)
    if size:
        out += b"\x28" + varint(size)
    if crc_masked:
        out += b"\x35" + struct.pack("<I", crc_masked)
    return out


def _parse_fields(buf):
    """Minimal protobuf wire parser -> list of (field, wire_type, value)."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = read_varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = read_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = read_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((f, wt, v))
    return out


def decode_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=0, slices=False)
    for f, wt, v in _parse_fields(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            for f2, _, dim in _parse_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, x in _parse_fields(dim):
                        if f3 == 1:
                            size = x
                    e["shape"].append(size)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
        elif f == 7:
            e["slices"] = True
    return e


def decode_header(buf):
    h = dict(num_shards=0, endianness=0, producer=0)
    for f, wt, v in _parse_fields(buf):
        if f == 1:
            h["num_shards"] = v
        elif f == 2:
            h["endianness"] = v
        elif f == 3:
            for f2, _, x in _parse_fields(v):
                if f2 == 1:
                    h["producer"] = x
    return h


# ----------------------------------------------------------------------------- table (SSTable) writer
class _BlockBuilder(object):
    def __init__(self, interval):
        self.interval = interval
        self.buf = bytearray()
        self.restarts = [0]
        self.counter = 0
        self.last = b""

    def add(self, key, value):
        shared = 0
        if self.counter < self.interval:
            n = min(len(self.last), len(key))
            while shared < n and self.last[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += varint(shared) + varint(len(key) - shared) + varint(len(value)) + key[shared:] + value
        self.last = key
        self.counter += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _shortest_separator(start, limit):
    n = min(len(start), len(limit))
    d = 0
    while d < n and start[d] == limit[d]:
        d += 1
    if d < n and start[d] < 0xFF and start[d] + 1 < limit[d]:
        return start[:d] + bytes([start[d] + 1])
    return start


def _short_successor(key):
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def build_table(items, block_size=BLOCK_SIZE):
    """items: sorted [(key bytes, value bytes)] -> the bytes of the .index file."""
    out = bytearray()
    index = _BlockBuilder(1)
    data = _BlockBuilder(RESTART_INTERVAL)
    pending = None  # (last key of the flushed block, handle)

    def write_block(contents):
        off = len(out)
        trailer_crc = mask_crc(crc32c(contents + b"\x00"))
        out.extend(contents + b"\x00" + struct.pack("<I", trailer_crc))
        return varint(off) + varint(len(contents))

    last_key = None
    for key, value in items:
        if last_key is not None and not key > last_key:
            raise ValueError("keys must be strictly increasing")
        if pending is not None:
            index.add(_shortest_separator(pending[0], key), pending[1])
            pending = None
        data.add(key, value)
        last_key = key
        if data.size() >= block_size:
            pending = (last_key, write_block(data.finish()))
            data = _BlockBuilder(RESTART_INTERVAL)
    if not data.empty():
        pending = (last_key, write_block(data.finish()))
    meta_handle = write_block(_BlockBuilder(RESTART_INTERVAL).finish())
    if pending is not None:
        index.add(_short_successor(pending[0]), pending[1])
    index_handle = write_block(index.finish())
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer))
    footer += struct.pack("<II", TABLE_MAGIC & 0xFFFFFFFF, TABLE_MAGIC >> 32)
    out.extend(footer)
    return bytes(out)


def _read_block(buf, off, size, verify=True):
    contents = buf[off:off + size]
    ctype = buf[off + size]
    stored = struct.unpack_from("<I", buf, off + size + 1)[0]
    if verify and unmask_crc(stored) != crc32c(bytes(contents) + bytes([ctype])):
        raise ValueError("table block at %d: checksum mismatch" % off)
    if ctype != 0:
        raise NotImplementedError("compressed table block (type %d); tensor bundles are written uncompressed" % ctype)
    nrest = struct.unpack_from("<I", contents, len(contents) - 4)[0]
    end = len(contents) - 4 - 4 * nrest
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = read_varint(contents, pos)
        non_shared, pos = read_varint(contents, pos)
        vlen, pos = read_varint(contents, pos)
        key = key[:shared] + bytes(contents[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(contents[pos:pos + vlen])))
        pos += vlen
    return out


def parse_table(buf, verify=True):
    """-> [(key, value)] of every entry of an SSTable held in `buf` (bytes)."""
    if len(buf) < 48:
        raise ValueError("not a table file: too short")
    lo, hi = struct.unpack_from("<II", buf, len(buf) - 8)
    if (hi << 32 | lo) != TABLE_MAGIC:
        raise ValueError("not a table file: bad magic %x" % (hi << 32 | lo))
    foot = buf[len(buf) - 48:len(buf) - 8]
    pos = 0
    _, pos = read_varint(foot, pos)
    _, pos = read_varint(foot, pos)
    ioff, pos = read_varint(foot, pos)
    isize, pos = read_varint(foot, pos)
    entries = []
    for _, handle in _read_block(buf, ioff, isize, verify):
        boff, p2 = read_varint(handle, 0)
        bsize, _ = read_varint(handle, p2)
        entries += _read_block(buf, boff, bsize, verify)
    return entries


# ----------------------------------------------------------------------------- bundle API
def data_filename(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def write_bundle(prefix, tensors, write_state=True):
    """tensors: {variable name: ndarray}.  Writes <prefix>.index, <prefix>.data-00000-of-00001 and,
    like Saver.save, the `checkpoint` state file next to them."""
    names = sorted(tensors, key=lambda s: s.encode())
    items = [(b"", encode_header(1))]
    offset = 0
    tmp = data_filename(prefix, 0, 1)
    with open(tmp + ".tmp", "wb") as f:
        for n in names:
            a = np.asarray(tensors[n])
            if not a.flags.c_contiguous:
                a = a.copy(order="C")  # (np.ascontiguousarray would turn a scalar into shape [1])
            if a.dtype not in DT:
                raise TypeError("%s: dtype %s not supported" % (n, a.dtype))
            if a.dtype.byteorder == ">":
                a = a.astype(a.dtype.newbyteorder("<"))
            f.write(a.tobytes())
            items.append((n.encode(), encode_entry(DT[a.dtype], a.shape, 0, offset, a.nbytes, mask_crc(crc32c(a)))))
            offset += a.nbytes
    os.replace(tmp + ".tmp", tmp)
    with open(prefix + ".index.tmp", "wb") as f:
        f.write(build_table(items))
    os.replace(prefix + ".index.tmp", prefix + ".index")
    if write_state:
        base = os.path.basename(prefix)
        with open(os.path.join(os.path.dirname(prefix) or ".", "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


def list_bundle(prefix):
    """-> (header dict, {name: entry dict}) from <prefix>.index."""
    with open(prefix + ".index", "rb") as f:
        buf = f.read()
    header, entries = None, {}
    for k, v in parse_table(buf):
        if k == b"":
            header = decode_header(v)
        else:
            entries[k.decode()] = decode_entry(v)
    if header is None:
        raise ValueError("%s.index: no bundle header" % prefix)
    if header["endianness"] != 0:
        raise NotImplementedError("big-endian bundle")
    return header, entries


def read_bundle(prefix, names=None, verify=True):
    """-> {variable name: ndarray}; `names` restricts the set (missing names raise KeyError)."""
    header, entries = list_bundle(prefix)
    want = list(entries) if names is None else list(names)
    files, out = {}, {}
    try:
        for n in want:
            e = entries[n]
            if e["slices"]:
                raise NotImplementedError("%s: partitioned (sliced) variable" % n)
            if e["dtype"] not in DT_INV:
                raise NotImplementedError("%s: tensor dtype enum %d" % (n, e["dtype"]))
            f = files.get(e["shard_id"])
            if f is None:
                f = files[e["shard_id"]] = open(data_filename(prefix, e["shard_id"], header["num_shards"]), "rb")
            f.seek(e["offset"])
            a = np.frombuffer(f.read(e["size"]), dtype=DT_INV[e["dtype"]].newbyteorder("<"))
            if a.nbytes != e["size"] or a.size != int(np.prod(e["shape"], dtype=np.int64)):
                raise ValueError("%s: size %d does not match shape %s" % (n, e["size"], e["shape"]))
            if verify and unmask_crc(e["crc32c"]) != crc32c(a):
                raise ValueError("%s: tensor checksum mismatch" % n)
            out[n] = a.reshape(e["shape"]).copy()
    finally:
        for f in files.values():
            f.close()
    return out


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by <directory>/checkpoint, or None."""
    p = os.path.join(directory, "checkpoint")
    if not os.path.exists(p):
        return None
    for line in open(p):
        if line.startswith("model_checkpoint_path:"):
            name = line.split(":", 1)[1].strip().strip('"')
            return name if os.path.isabs(name) else os.path.join(directory, name)
    return None
