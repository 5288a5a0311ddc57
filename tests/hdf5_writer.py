"""An independent WRITER of the HDF5 structures `vae_captioning_amd/utils/hdf5_min.py` reads -- test infrastructure, written from the
format specification (version 3.0, sections III.A-E, IV.A.1-2), byte layouts spelled out here on purpose rather than shared with the
reader: superblock version 0, a root group held by a symbol table (B-tree version 1 with an optional second level, symbol nodes,
local heap), data sets with version-1 object headers (dataspace version 1, fixed-point / float datatype, fill-value message that a
reader must skip, data layout version 3 contiguous -- or chunked / compact, to check the refusals), an optional user block and an
optional continuation block in the data set's object header.  Mirrors what `h5py.File(p, "w").create_dataset(name, shape, dtype)`
produces with the default `libver='earliest'` (preprocess.py:25-45); h5py itself is absent from this image."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(typ, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", typ, len(data), flags) + data


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind in "iu":
        bits = (8 if dt.kind == "i" else 0) | (1 if dt.byteorder == ">" else 0)
        return struct.pack("<BBBBI", 0x10 | 0, bits, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    if dt.kind == "f":
        props = {4: struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127), 8: struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)}[dt.itemsize]
        return struct.pack("<BBBBI", 0x10 | 1, 0x20 | (1 if dt.byteorder == ">" else 0), 0x3f if dt.itemsize == 8 else 0x1f, 0, dt.itemsize) + props
    raise ValueError(dt)


def write(path, datasets, user_block=0, two_level_btree=False, layout="contiguous", continuation=False):
    """datasets: {name: array or (shape, dtype) for a never-written (unallocated) data set}."""
    out = bytearray(b"\0" * user_block)
    base = user_block
    place = lambda b: (out.extend(b"\0" * (-len(out) % 8)), len(out) - base, out.extend(b))[1]   # -> address relative to the base
    out.extend(b"\0" * 96)   # superblock, filled in last
    names = sorted(datasets)   # (a group B-tree is ordered by name)
    heap_data = bytearray(b"\0" * 8)
    name_off = {}
    for n in names:
        name_off[n] = len(heap_data)
        heap_data.extend(_pad8(n.encode() + b"\0"))
    # raw data + data-set object headers
    headers = {}
    for n in names:
        d = datasets[n]
        if isinstance(d, tuple):
            shape, dt, addr, nbytes = d[0], np.dtype(d[1]), UNDEF, int(np.prod(d[0])) * np.dtype(d[1]).itemsize
        else:
            a = np.ascontiguousarray(d)
            shape, dt, nbytes = a.shape, a.dtype, a.nbytes
            addr = place(a.tobytes())
        space = struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", s) for s in shape)
        if layout == "contiguous":
            lay = struct.pack("<BB", 3, 1) + struct.pack("<QQ", addr, nbytes)
        elif layout == "chunked":
            lay = struct.pack("<BBB", 3, 2, len(shape) + 1) + struct.pack("<Q", UNDEF) + b"".join(struct.pack("<I", 1) for _ in range(len(shape) + 1))
        else:
            lay = struct.pack("<BBH", 3, 0, 0)
        fill = struct.pack("<BBBB", 2, 2, 0, 0)                      # fill value message, version 2, undefined: to be skipped
        mtime = struct.pack("<B3xI", 1, 0)                            # modification time message: to be skipped
        first = [_msg(0x0001, space), _msg(0x0005, fill), _msg(0x0003, _dtype_msg(dt))]
        rest = [_msg(0x0008, lay), _msg(0x0012, mtime)]
        if continuation:
            caddr = place(b"".join(rest))
            body = b"".join(first) + _msg(0x0010, struct.pack("<QQ", caddr, len(b"".join(rest))))
            nmsg = len(first) + 1 + len(rest)
        else:
            body = b"".join(first + rest)
            nmsg = len(first) + len(rest)
        headers[n] = place(struct.pack("<BxHII4x", 1, nmsg, 1, len(body)) + body)
    heap_addr_data = place(bytes(heap_data))
    heap = place(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), UNDEF, heap_addr_data))
    # symbol nodes: all links in one node, or one node per link under a level-1 B-tree node
    def snod(ns):
        ent = b"".join(struct.pack("<QQII16x", name_off[n], headers[n], 0, 0) for n in ns)
        return place(b"SNOD" + struct.pack("<BxH", 1, len(ns)) + ent)
    def tree(level, children, keys):
        b = b"TREE" + struct.pack("<BBHQQ", 0, level, len(children), UNDEF, UNDEF) + struct.pack("<Q", keys[0])
        for c, k in zip(children, keys[1:]):
            b += struct.pack("<QQ", c, k)
        return place(b)
    if two_level_btree and len(names) > 1:
        leaves = [tree(0, [snod([n])], [0 if i == 0 else name_off[names[i - 1]], name_off[n]]) for i, n in enumerate(names)]
        root_bt = tree(1, leaves, [0] + [name_off[n] for n in names])
    else:
        root_bt = tree(0, [snod(names)], [0, name_off[names[-1]] if names else 0])
    root_hdr_body = _msg(0x0011, struct.pack("<QQ", root_bt, heap))
    root_hdr = place(struct.pack("<BxHII4x", 1, 1, 1, len(root_hdr_body)) + root_hdr_body)
    sb = (b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0) +
          struct.pack("<QQQQ", 0 if not user_block else user_block, UNDEF, len(out) - base, UNDEF) +
          struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", root_bt, heap))
    assert len(sb) == 96
    out[user_block:user_block + 96] = sb
    with open(path, "wb") as fh:
        fh.write(bytes(out))
