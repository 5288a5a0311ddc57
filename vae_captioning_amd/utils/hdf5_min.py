"""Minimal HDF5 READER for the one file the reference keeps in that format: `images (N, 224, 224, 3) uint8`, written by
`preprocess.py:25-45` with `h5py.File(h5_file, "w").create_dataset("images", shape, dtype='uint8')` and read by
`utils/batch_gen.py:35-41,189` as `h5py.File(hdf5_file, 'r')['images'][sorted indices]`.

h5py is not part of this image, so the real-data `--fine_tune` path could not open the reference's own file.  What h5py writes for that
call (default `libver='earliest'`, no chunking, no filters) is the oldest and simplest layout of the format, which this module reads
from the published specification ("HDF5 File Format Specification Version 3.0", sections III.A-E, IV.A.1-2):

    superblock version 0 / 1  ->  root group symbol-table entry  ->  object header (version 1) of the root group with a Symbol Table
    message (0x0011: B-tree + local heap)  ->  B-tree version 1 group nodes ("TREE") -> symbol nodes ("SNOD") whose entries name their
    links through the local heap ("HEAP")  ->  the data set's object header: Dataspace (0x0001), Datatype (0x0003: fixed-point or IEEE
    float), Data Layout (0x0008, version 3 contiguous: address + size; versions 1 / 2 contiguous too)

and memory-maps the contiguous raw data.  Anything else the format allows -- chunked or compact storage, filters, version-2 object
headers ("OHDR": `libver='latest'`), superblock versions 2 / 3, variable-length types -- raises NotImplementedError naming what was met.
Status: FORMAT-SPEC READER.  It is exercised against an independent writer of the same structures in
`tests/test_hdf5_min.py`; no file written by h5py / libhdf5 has ever been read here (neither exists in this image)."""
import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5FormatError(ValueError):
    pass


class Dataset(object):
    """`f['images']`: shape, dtype and numpy-style reads over a memory map of the contiguous raw data.  Index lists are read in the
    order given (the reference sorts them first, utils/batch_gen.py:347-362: h5py requires increasing order; this reader does not)."""

    def __init__(self, mm, shape, dtype):
        self._a, self.shape, self.dtype = mm, tuple(shape), np.dtype(dtype)

    def __len__(self):
        return self.shape[0]

    @property
    def ndim(self):
        return len(self.shape)

    def __getitem__(self, key):
        if isinstance(key, (list, tuple)) and key and not isinstance(key[0], slice) and all(isinstance(k, (int, np.integer)) for k in key):
            key = np.asarray(key, np.int64)
        return np.asarray(self._a[key])

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self._a)
        return a.astype(dtype) if dtype is not None else a


class File(object):
    """Read-only.  `File(path)['images']` -> Dataset; `keys()` lists the root group's links; usable as a context manager."""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise NotImplementedError("hdf5_min reads only (this build's preprocess.py writes the image array as .npy)")
        self.path = path
        self._fh = open(path, "rb")
        self._links = None
        self.base = 0            # (absolute offsets until the superblock has told the base address)
        self._parse_superblock()

    # ---- plumbing
    def _read(self, off, n):
        self._fh.seek(self.base + off)
        b = self._fh.read(n)
        if len(b) != n:
            raise Hdf5FormatError("%s: truncated file (wanted %d bytes at offset %d)" % (self.path, n, off))
        return b

    def _addr(self, b, o):
        return int.from_bytes(b[o:o + self.so], "little")

    def _len(self, b, o):
        return int.from_bytes(b[o:o + self.sl], "little")

    def close(self):
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- III.A superblock (versions 0 and 1), found at offset 0, 512, 1024, ...
    def _parse_superblock(self):
        off = 0
        self._fh.seek(0, 2)
        end = self._fh.tell()
        while True:
            self._fh.seek(off)
            if self._fh.read(8) == SIGNATURE:
                break
            off = 512 if off == 0 else off * 2
            if off >= end:
                raise Hdf5FormatError("%s: no HDF5 signature" % self.path)
        self._fh.seek(off)
        sb = self._fh.read(128)
        ver = sb[8]
        if ver not in (0, 1):
            raise NotImplementedError("%s: superblock version %d (libver='latest' files): only the versions h5py's default writes (0, 1) are read" % (self.path, ver))
        self.so, self.sl = sb[13], sb[14]
        if self.so not in (4, 8) or self.sl not in (4, 8):
            raise Hdf5FormatError("%s: size of offsets / lengths %d / %d" % (self.path, self.so, self.sl))
        o = 24 if ver == 0 else 28   # version 1 adds the indexed-storage K (2 bytes) + 2 reserved
        base = self._addr(sb, o)
        o += 4 * self.so             # base address, free-space info, end of file, driver information block
        # root group symbol table entry: link name offset, object header address, cache type, reserved, 16 bytes of scratch
        self.root_header = self._addr(sb, o + self.so)
        cache = struct.unpack_from("<I", sb, o + 2 * self.so)[0]
        scratch = o + 2 * self.so + 8
        self.root_cached = (self._addr(sb, scratch), self._addr(sb, scratch + self.so)) if cache == 1 else None
        self.base = base + off if base == 0 else base   # (a user block shifts the file: addresses are relative to the base address)

    # ---- IV.A.1 version-1 object header: every message (type, data), continuation blocks followed
    def _messages(self, addr):
        h = self._read(addr, 16)
        if h[:4] == b"OHDR":
            raise NotImplementedError("%s: version-2 object header (a file written with libver='latest')" % self.path)
        if h[0] != 1:
            raise Hdf5FormatError("%s: object header version %d at %d" % (self.path, h[0], addr))
        nmsg, = struct.unpack_from("<H", h, 2)
        size, = struct.unpack_from("<I", h, 8)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            boff, blen = blocks.pop(0)
            blk = self._read(boff, blen)
            p = 0
            while p + 8 <= blen and len(out) < nmsg:
                typ, sz, flags = struct.unpack_from("<HHB", blk, p)
                data = blk[p + 8:p + 8 + sz]
                p += 8 + sz
                if typ == 0x0010:    # continuation: offset, length
                    blocks.append((self._addr(data, 0), self._len(data, self.so)))
                out.append((typ, data))
        return out

    # ---- III.B-E old-style group: Symbol Table message -> B-tree of symbol nodes, names in the local heap
    def _heap_data(self, addr):
        h = self._read(addr, 8 + 2 * self.sl + self.so)
        if h[:4] != b"HEAP":
            raise Hdf5FormatError("%s: no local heap at %d" % (self.path, addr))
        return self._addr(h, 8 + 2 * self.sl), self._len(h, 8)

    def _name(self, heap, off):
        data_addr, size = heap
        raw = self._read(data_addr + off, min(256, size - off))
        return raw.split(b"\0", 1)[0].decode("utf-8")

    def _walk_btree(self, addr, heap, links):
        h = self._read(addr, 8 + 2 * self.so)
        if h[:4] == b"SNOD":
            n, = struct.unpack_from("<H", h, 6)
            esz = 2 * self.so + 24
            body = self._read(addr + 8, n * esz)
            for i in range(n):
                e = body[i * esz:(i + 1) * esz]
                links[self._name(heap, self._addr(e, 0))] = self._addr(e, self.so)
            return
        if h[:4] != b"TREE":
            raise Hdf5FormatError("%s: neither a B-tree node nor a symbol node at %d" % (self.path, addr))
        if h[4] != 0:
            raise Hdf5FormatError("%s: B-tree node type %d in a group" % (self.path, h[4]))
        used, = struct.unpack_from("<H", h, 6)
        body = self._read(addr + 8 + 2 * self.so, (used + 1) * self.sl + used * self.so)
        p = self.sl    # key 0
        for _ in range(used):
            self._walk_btree(self._addr(body, p), heap, links)
            p += self.so + self.sl
        return

    def _root_links(self):
        if self._links is None:
            bt = None
            for typ, data in self._messages(self.root_header):
                if typ == 0x0011:
                    bt = (self._addr(data, 0), self._addr(data, self.so))
                elif typ in (0x0002, 0x0006):
                    raise NotImplementedError("%s: new-style group (link messages): a file written with libver='latest'" % self.path)
            if bt is None:
                bt = self.root_cached
            if bt is None:
                raise Hdf5FormatError("%s: the root group has no symbol table" % self.path)
            links = {}
            self._walk_btree(bt[0], self._heap_data(bt[1]), links)
            self._links = links
        return self._links

    def keys(self):
        return sorted(self._root_links())

    def __contains__(self, name):
        return name.strip("/") in self._root_links()

    # ---- the data set: dataspace, datatype, contiguous layout
    def __getitem__(self, name):
        links = self._root_links()
        name = name.strip("/")
        if name not in links:
            raise KeyError("%s: no object %r in the root group (has: %s)" % (self.path, name, ", ".join(sorted(links)) or "nothing"))
        shape = dtype = layout = None
        for typ, d in self._messages(links[name]):
            if typ == 0x0001:     # dataspace: version, rank, flags, (v1: 5 reserved bytes | v2: type), dimensions
                ver, rank = d[0], d[1]
                o = 8 if ver == 1 else 4
                shape = tuple(self._len(d, o + i * self.sl) for i in range(rank))
            elif typ == 0x0003:   # datatype: class | version << 4, three bytes of class bits, size
                cls, bits0 = d[0] & 15, d[1]
                size, = struct.unpack_from("<I", d, 4)
                order = ">" if (bits0 & 1) else "<"
                if cls == 0:
                    dtype = np.dtype("%s%s%d" % (order, "i" if (bits0 & 8) else "u", size))
                elif cls == 1:
                    dtype = np.dtype("%sf%d" % (order, size))
                else:
                    raise NotImplementedError("%s: %r has datatype class %d (only fixed-point and floating-point are read)" % (self.path, name, cls))
            elif typ == 0x0008:   # data layout
                ver = d[0]
                if ver == 3:
                    if d[1] != 1:
                        raise NotImplementedError("%s: %r is stored %s (only contiguous storage is read: what create_dataset writes without chunks)"
                                                  % (self.path, name, {0: "compact", 2: "chunked"}.get(d[1], "in layout class %d" % d[1])))
                    layout = (self._addr(d, 2), self._len(d, 2 + self.so))
                elif ver in (1, 2):
                    rank, cls = d[1], d[2]
                    if cls != 1:
                        raise NotImplementedError("%s: %r: layout class %d (only contiguous storage is read)" % (self.path, name, cls))
                    layout = (self._addr(d, 8), None)
                else:
                    raise NotImplementedError("%s: %r: data layout message version %d" % (self.path, name, ver))
            elif typ == 0x000B:
                raise NotImplementedError("%s: %r has a filter pipeline (compressed / chunked data)" % (self.path, name))
        if shape is None or dtype is None or layout is None:
            raise Hdf5FormatError("%s: %r is not a data set (dataspace / datatype / layout message missing)" % (self.path, name))
        n = int(np.prod(shape)) if shape else 1
        if layout[0] == UNDEF >> (64 - 8 * self.so):   # space never allocated: nothing was ever written (late allocation): fill value 0
            return Dataset(np.zeros(shape, dtype), shape, dtype)
        if layout[1] is not None and layout[1] < n * dtype.itemsize:
            raise Hdf5FormatError("%s: %r: layout size %d < %d" % (self.path, name, layout[1], n * dtype.itemsize))
        mm = np.memmap(self.path, mode="r", dtype=dtype, offset=self.base + layout[0], shape=shape)
        return Dataset(mm, shape, dtype)
