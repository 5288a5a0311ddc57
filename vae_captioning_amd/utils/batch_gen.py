"""Batch assembly with the tensor contract of the reference's Batch_Generator
(utils/batch_gen.py:164-205, 296-345) followed by preprocess_captions
(utils/caption_utils.py:4-25): what main.py:226-238 finally feeds.

`BatchGenerator` is the compact in-memory form (captions + feature dict); `Batch_Generator` further down
keeps the reference class's constructor and its train / val / test generators over image directories.
"""
import glob as _glob
import json as _json
import pickle as _pickle

import numpy as np


def form_captions_batch(indexed, names, num_captions=1, rng=None):
    """indexed: {file_name: [[ids of <BOS> w.. <EOS>], ...]}.  Returns (inputs, labels, lengths):
    inputs = caption[:-1], labels = caption[1:], lengths = len(caption) - 1; images with fewer than
    num_captions captions leave all-PAD slots of length 0; everything is padded with 0 to the longest
    caption of the batch.  num_captions == 1 picks ONE caption at random (random_select)."""
    nb = len(names)
    ins = [[[0] for _ in range(num_captions)] for _ in range(nb)]
    lab = [[[0] for _ in range(num_captions)] for _ in range(nb)]
    lengths = np.zeros((nb, num_captions), np.int32)
    for i, fn in enumerate(names):
        caps = indexed[fn.split("/")[-1]]
        if num_captions == 1 and len(caps):
            r = rng if rng is not None else np.random
            caps = [caps[int(r.integers(len(caps))) if hasattr(r, "integers") else r.randint(len(caps))]]
        for k, cap in enumerate(caps[:num_captions]):
            ins[i][k], lab[i][k], lengths[i][k] = cap[:-1], cap[1:], len(cap) - 1
    pad = max(len(c) for row in ins for c in row)
    to_arr = lambda rows: np.array([[c + [0] * (pad - len(c)) for c in row] for row in rows], np.int32)
    return to_arr(ins), to_arr(lab), lengths


def preprocess_captions(inputs, labels, lengths, c_v=None):
    """[B, nc, T] -> [B*nc, T]; cluster vectors repeated per caption (utils/caption_utils.py:4-25)."""
    B, nc, T = inputs.shape
    out = dict(cap_dec=inputs.reshape(B * nc, T), cap_enc=labels.reshape(B * nc, T), lengths=lengths.reshape(-1))
    if c_v is not None and len(c_v):
        out["c_v"] = np.repeat(np.asarray(c_v, np.float32), nc, axis=0)
    return out


def shard_ranges(n, batch_size, shard=None):
    """[lo, hi) index ranges of the successive batches one rank reads from a list of n shuffled examples.
    shard None: the reference's chunks of batch_size with a ragged last one (utils/batch_gen.py:180-183).
    shard (rank, world): data-parallel training -- the list is cut into GLOBAL batches of world*batch_size examples and
    rank r reads rows [r*batch_size, (r+1)*batch_size) of each (dp.shard_batch's rule); a ragged last global batch is
    dropped on every rank alike (shapes are static on device and the ranks must step together)."""
    if shard is None:
        return [(s, min(s + batch_size, n)) for s in range(0, n, batch_size)]
    rank, world = shard
    g = batch_size * world
    if n % g and rank == 0 and n >= g:
        print("data-parallel epoch: the ragged last global batch (%d of %d examples) is dropped on every rank" % (n % g, n))
    return [(s + rank * batch_size, s + (rank + 1) * batch_size) for s in range(0, n - g + 1, g)]


def global_range(lo, batch_size, shard):
    """[glo, ghi) of the GLOBAL batch that rank-local range [lo, lo + batch_size) belongs to.  Data-parallel ranks form the captions
    of the whole global batch (host-side list work) and keep their rows: every rank then pads to the SAME length T (the longest
    caption of the global batch -- per-rank padding would give the ranks different kernel shapes and step times, and they meet
    at the gradient all-reduce) and consumes the shared random stream identically (random_select draws)."""
    rank, world = shard
    glo = lo - rank * batch_size
    return glo, glo + batch_size * world


class BatchGenerator(object):
    """next_batch(): dicts in the layout Trainer.set_batch takes.  features: {file_name: [1, F] or [F]}
    (the reference's feature pickles); cluster vectors: {file_name: 91-vector}, column 0 dropped
    (main.py:236)."""

    def __init__(self, indexed_captions, features, batch_size, cluster_vectors=None, seed=42, shard=None):
        self.caps, self.feats, self.bs, self.cv = indexed_captions, features, batch_size, cluster_vectors
        self.names = [n for n in indexed_captions if n in features]
        self.rng = np.random.default_rng(seed)
        self.shard = shard  # (rank, world): every rank shuffles identically and takes its slice of each GLOBAL batch

    def next_batch(self, use_obj_vectors=False, num_captions=1, shuffle=True):
        order = self.rng.permutation(len(self.names)) if shuffle else np.arange(len(self.names))
        for lo, hi in shard_ranges(len(order), self.bs, self.shard):
            names = [self.names[i] for i in order[lo:hi]]
            if self.shard is not None:   # captions of the GLOBAL batch, this rank's rows (same T and random draws on every rank)
                glo, ghi = global_range(lo, self.bs, self.shard)
                ins, lab, lens = (a[lo - glo:hi - glo] for a in form_captions_batch(self.caps, [self.names[i] for i in order[glo:ghi]], num_captions, self.rng))
            else:
                ins, lab, lens = form_captions_batch(self.caps, names, num_captions, self.rng)
            if ins.ndim == 2:
                ins, lab, lens = ins[:, None, :], lab[:, None, :], lens.reshape(-1, 1)
            cv = None
            if use_obj_vectors and self.cv is not None:
                cv = np.stack([np.asarray(self.cv[n], np.float32)[1:] for n in names])
            b = preprocess_captions(ins, lab, lens, cv)
            b["features"] = np.stack([np.asarray(self.feats[n], np.float32).reshape(-1) for n in names])
            b["names"] = names
            yield b


# ------------------------------------------------------------------------------------------------
# The reference's generator class, file-system facing (utils/batch_gen.py:16-369): image names come
# from a directory listing, images from precomputed feature dicts, from a preprocessed image array
# (the HDF5 file's role) or from the image files themselves.
# ------------------------------------------------------------------------------------------------
def open_image_array(path):
    """The `images (N, 224, 224, 3) uint8` array preprocess.py writes.  The reference keeps it in HDF5 (preprocess.py:25-45) and reads it
    with h5py (utils/batch_gen.py:35-41); without h5py (this image) the file is opened by `hdf5_min`, a reader of exactly the layout
    that call writes (superblock 0, old-style root group, one contiguous data set); this build's own preprocess.py writes the same
    array as .npy, which is memory-mapped."""
    if path.endswith(".npy"):
        return np.load(path, mmap_mode="r")
    try:
        import h5py
    except ImportError:
        from . import hdf5_min
        return hdf5_min.File(path)["images"]
    return h5py.File(path, "r")["images"]


class Batch_Generator(object):
    """Constructor arguments, attributes and generator methods of utils/batch_gen.py:16-369.
    next_batch      -> images, (inputs, labels), lengths, c_v           (training, shuffled, ragged last batch)
    next_val_batch  -> images, (inputs, labels), lengths[, image_ids], c_v
    next_test_batch -> images, image_ids, c_v
    `images` is [B, 4096] when a feature dict is given, else [B, 224, 224, 3] uint8."""

    def __init__(self, train_dir, train_cap_json=None, captions=None, batch_size=None, use_hdf5=False, hdf5_file=None,
                 feature_dict=None, get_image_ids=False, get_test_ids=False, val_tr_unused=None,
                 cluster_vectors=None, seed=42, shard=None):
        self.shard = shard  # (rank, world) for data-parallel training (see shard_ranges); None = the reference's behaviour
        self.use_hdf5 = bool(use_hdf5) and feature_dict is None
        if self.use_hdf5:
            if not hdf5_file:
                raise ValueError("Specify hdf5 file path")
            import os
            npy = os.path.splitext(hdf5_file)[0] + ".npy"
            if not os.path.exists(hdf5_file) and os.path.exists(npy):
                hdf5_file = npy  # what this build's preprocess.py writes
            if not os.path.exists(hdf5_file):
                print("No preprocessed image array at %s: decoding the image files per batch" % hdf5_file)
                self.use_hdf5 = False
        if self.use_hdf5:
            with open("./pickles/itoi.pickle", "rb") as rf:
                self.imtoi = _pickle.load(rf)
            self.images = open_image_array(hdf5_file)
        self._iterable = sorted(_glob.glob(train_dir + "*.jpg")) if val_tr_unused is None else list(val_tr_unused)
        self._train_dir = train_dir
        self._batch_size = batch_size or len(self._iterable)
        if len(self._iterable) == 0:
            raise FileNotFoundError("no *.jpg under %s" % train_dir)
        self._train_cap_json = train_cap_json
        if get_test_ids:  # the test set has no captions: ids come from image_info_test2014.json
            with open(train_cap_json) as rf:
                self._fn_to_id = {img["file_name"]: img["id"] for img in _json.load(rf)["images"]}
        self.cap_instance = captions
        self.captions = captions.captions_indexed if captions is not None else None
        self.val_cap_instance, self.val_captions, self.val_feature_dict = None, None, None
        self.rng = np.random.RandomState(seed)  # the reference seeds numpy globally with 42 here (:65-66)
        self.feature_dict = feature_dict
        self.get_image_ids = get_image_ids
        self.unused_cap_in = None
        self._cluster_vectors = cluster_vectors  # {file_name: 91-vector}; default: ./obj_vectors/c_v*.pickle

    # -- training on train + part of val (utils/batch_gen.py:71-96)
    def repartiton(self, val_cap_instance, val_feature_dict, gen_val_cap):
        if not val_cap_instance:
            raise ValueError("If use validation set images for training need to specify val_cap instance")
        if not val_feature_dict:
            raise ValueError("If use validation set images for training need to specify val_feature_dict")
        self.val_cap_instance, self.val_captions = val_cap_instance, val_cap_instance.captions_indexed
        val_dir = "/".join(self._train_dir.split("/")[:-2] + ["val2014/"])
        val_list = sorted(_glob.glob(val_dir + "*.jpg"))
        self.rng.shuffle(val_list)
        if gen_val_cap is not None and gen_val_cap > 0:
            self._iterable.extend(val_list[:-gen_val_cap])
            self.unused_cap_in = val_list[-gen_val_cap:]
        else:
            self._iterable.extend(val_list)
        self.val_feature_dict = val_feature_dict

    # -- pieces
    def _cv_dict(self, load_test=False):
        if self._cluster_vectors is not None:
            return self._cluster_vectors
        with open("./obj_vectors/c_v_test.pickle" if load_test else "./obj_vectors/c_v.pickle", "rb") as rf:
            c_v = _pickle.load(rf)
        assert isinstance(c_v, dict), "cluster vector pickle must contain dict"
        return c_v

    def _lookup(self, d, alt, key):
        if key in d:
            return d[key]
        if alt is not None and key in alt:
            return alt[key]
        raise KeyError(key)

    def _images_c_v(self, names, c_v):
        base = [n.split("/")[-1] for n in names]
        cl_v = np.array([np.asarray(c_v.get(b, np.zeros(91)), np.float64) for b in base]) if c_v else np.array([])
        if self.feature_dict:
            images = np.stack([np.asarray(self._lookup(self.feature_dict, self.val_feature_dict, b)).reshape(-1) for b in base])
        elif self.use_hdf5:
            images = np.asarray(self.images[[self.imtoi[b] for b in base]])
        else:
            from .image_utils import load_image
            images = np.stack([load_image(n) for n in names])
        return images, cl_v

    def _sorted_for_array(self, names):
        """HDF5 fancy indexing wants increasing indices: reorder the batch by (index, name) (:152-162)."""
        if not self.use_hdf5:
            return list(names)
        return [n for _, n in sorted((self.imtoi[n.split("/")[-1]], n) for n in names)]

    def _captions_for(self, names, random_select=True, num_captions=1):
        indexed = {}
        for n in names:
            b = n.split("/")[-1]
            caps = self.captions.get(b) or (self.val_captions or {}).get(b) or []
            indexed[b] = caps
        nc = 1 if random_select else num_captions
        ins, lab, lens = form_captions_batch(indexed, names, nc, self.rng)
        if nc == 1:  # "if using random_select, temporary" squeeze (:341-344)
            return ins[:, 0], lab[:, 0], lens[:, 0]
        return ins, lab, lens

    def _chunks(self, names, shard=None):
        for lo, hi in shard_ranges(len(names), self._batch_size, shard):
            yield self._sorted_for_array(names[lo:hi])

    def _chunks_global(self, names, shard):
        """data-parallel: (this rank's names, the names of the whole global batch in rank order, this rank's row range in it)"""
        rank, world = shard
        bs = self._batch_size
        for lo, hi in shard_ranges(len(names), bs, shard):
            glo, _ = global_range(lo, bs, shard)
            per_rank = [self._sorted_for_array(names[glo + r * bs:glo + (r + 1) * bs]) for r in range(world)]
            yield per_rank[rank], [n for pr in per_rank for n in pr], (rank * bs, (rank + 1) * bs)

    def _imid(self, names, test=False):
        if test:
            return [self._fn_to_id[n.split("/")[-1]] for n in names]
        alt = self.val_cap_instance.filename_to_imid if self.val_cap_instance is not None else None
        return [self._lookup(self.cap_instance.filename_to_imid, alt, n.split("/")[-1]) for n in names]

    # -- generators
    def next_batch(self, use_obj_vectors=False, num_captions=1):
        c_v = self._cv_dict() if use_obj_vectors else None
        self.rng.shuffle(self._iterable)  # same seed on every rank: identical order, disjoint slices
        if self.shard is not None:
            for names, gnames, (r0, r1) in self._chunks_global(self._iterable, self.shard):
                images, cl_v = self._images_c_v(names, c_v)
                ins, lab, lens = (a[r0:r1] for a in self._captions_for(gnames, num_captions == 1, num_captions))   # T of the GLOBAL batch
                yield images, (ins, lab), lens, cl_v
            return
        for names in self._chunks(self._iterable, self.shard):
            images, cl_v = self._images_c_v(names, c_v)
            ins, lab, lens = self._captions_for(names, num_captions == 1, num_captions)
            yield images, (ins, lab), lens, cl_v

    def next_val_batch(self, get_image_ids=False, use_obj_vectors=False):
        self.get_image_ids = get_image_ids
        c_v = self._cv_dict() if use_obj_vectors else None
        for names in self._chunks(self._iterable):
            images, cl_v = self._images_c_v(names, c_v)
            ins, lab, lens = self._captions_for(names)
            if get_image_ids:
                yield images, (ins, lab), lens, self._imid(names), cl_v
            else:
                yield images, (ins, lab), lens, cl_v

    def next_test_batch(self, use_obj_vectors=False):
        c_v = self._cv_dict(True) if use_obj_vectors else None
        for names in self._chunks(self._iterable):
            images, cl_v = self._images_c_v(names, c_v)
            yield images, self._imid(names, True), cl_v

    @property
    def cap_dict(self):
        return self.captions

    def set_bs(self, batch_size):
        self._batch_size = batch_size


def feed_dict(images, captions, lengths, c_v, num_captions, fine_tune):
    """One generator item -> the dict Trainer.set_batch takes (main.py:226-238: preprocess_captions when
    num_captions > 1, cluster-vector column 0 dropped)."""
    ins, lab = captions
    if ins.ndim == 2:
        ins, lab, lengths = ins[:, None, :], lab[:, None, :], np.asarray(lengths).reshape(-1, 1)
    cv = np.asarray(c_v, np.float32)[:, 1:] if c_v is not None and len(c_v) else None
    b = preprocess_captions(ins.astype(np.int32), lab.astype(np.int32), np.asarray(lengths, np.int32), cv)
    # fine-tuning: uint8 pixels stay uint8 (the device casts them, Trainer.set_batch); precomputed features are float32
    b["images" if fine_tune else "features"] = np.asarray(images) if (fine_tune and np.asarray(images).dtype == np.uint8) else np.asarray(images, np.float32)
    if fine_tune:
        b["features"] = np.zeros((len(images), 0), np.float32)
    return b
