"""CPU: tokeniser / vocabulary / batch contract of the data layer (utils/captions.py,
utils/batch_gen.py:296-345, utils/caption_utils.py) on a hand-made COCO-style annotation set with
known answers worked out from the reference's rules."""
import numpy as np

from vae_captioning_amd.utils.batch_gen import BatchGenerator, form_captions_batch, preprocess_captions
from vae_captioning_amd.utils.captions import Captions, Dictionary, tokenize

COCO = {
    "images": [{"id": 7, "file_name": "a.jpg"}, {"id": 9, "file_name": "b.jpg"}, {"id": 11, "file_name": "c.jpg"}],
    "annotations": [
        {"image_id": 7, "caption": "A man riding a wave."},
        {"image_id": 7, "caption": "a man, on a  surf-board!"},
        {"image_id": 9, "caption": "A dog runs"},
        {"image_id": 9, "caption": "a dog"},
        {"image_id": 9, "caption": "the man and a dog"},
        {"image_id": 11, "caption": "A zebra"},
    ],
}


def test_tokenizer_rules():
    assert tokenize("A man riding a wave.") == ["<BOS>", "a", "man", "riding", "a", "wave", "<EOS>"]
    assert tokenize("a man, on a  surf-board!") == ["<BOS>", "a", "man", "on", "a", "surf", "board", "<EOS>"]
    long = " ".join(["word"] * 150)
    assert len(tokenize(long)) == 152  # cap_max_length never clips (utils/captions.py:32-34)


def test_vocabulary_order_and_ids():
    caps = Captions(COCO)
    d = Dictionary(caps.captions, keep_words=2)
    # counts: a 8, <BOS> 6, <EOS> 6, man 3, dog 3, everything else 1 (< 2, dropped) and <UNK> (kept, count 1)
    assert d.word2idx["<PAD>"] == 0 and d.idx2word[0] == "<PAD>"
    assert [d.idx2word[i] for i in range(1, 7)] == ["a", "<BOS>", "<EOS>", "dog", "man", "<UNK>"]
    assert d.vocab_size == 7 and len(d) == 7
    idx = caps.index_captions(d.word2idx)
    assert idx["c.jpg"] == [[2, 1, 6, 3]]  # 'zebra' -> <UNK>
    assert caps.filename_to_imid["b.jpg"] == 9


def test_batch_contract():
    caps = Captions(COCO)
    d = Dictionary(caps.captions, keep_words=1)
    idx = caps.index_captions(d.word2idx)
    ins, lab, lens = form_captions_batch(idx, ["x/a.jpg", "b.jpg", "c.jpg"], num_captions=3)
    assert ins.shape == lab.shape == (3, 3, 7) and lens.shape == (3, 3)
    bos, eos = d.word2idx["<BOS>"], d.word2idx["<EOS>"]
    assert np.all(ins[:, :, 0][lens > 0] == bos)
    for i in range(3):
        for k in range(3):
            L = lens[i, k]
            if L == 0:  # image with fewer captions: all-PAD row (batch_gen.py:313-317)
                assert not ins[i, k].any() and not lab[i, k].any()
            else:
                assert lab[i, k, L - 1] == eos and not lab[i, k, L:].any() and not ins[i, k, L:].any()
                assert list(ins[i, k, 1:L]) == list(lab[i, k, :L - 1])
    assert lens.tolist() == [[6, 7, 0], [4, 3, 6], [3, 0, 0]]
    cv = np.arange(3 * 90, dtype=np.float32).reshape(3, 90)
    b = preprocess_captions(ins, lab, lens, cv)
    assert b["cap_dec"].shape == (9, 7) and b["lengths"].tolist() == [6, 7, 0, 4, 3, 6, 3, 0, 0]
    np.testing.assert_array_equal(b["c_v"][3:6], np.repeat(cv[1:2], 3, axis=0))


def test_generator_feeds_trainer_layout():
    caps = Captions(COCO)
    d = Dictionary(caps.captions, keep_words=1)
    idx = caps.index_captions(d.word2idx)
    feats = {n: np.full((1, 8), i, np.float32) for i, n in enumerate(idx)}
    cvs = {n: np.arange(91, dtype=np.float32) for n in idx}
    g = BatchGenerator(idx, feats, batch_size=2, cluster_vectors=cvs, seed=1)
    batches = list(g.next_batch(use_obj_vectors=True, num_captions=2))
    assert [b["features"].shape[0] for b in batches] == [2, 1]
    b = batches[0]
    assert b["cap_dec"].shape[0] == 4 and b["c_v"].shape == (4, 90) and b["c_v"][0, 0] == 1.0  # column 0 dropped (main.py:236)
    one = list(BatchGenerator(idx, feats, 3, seed=2).next_batch(num_captions=1))[0]
    assert one["cap_dec"].shape[0] == 3 and (one["lengths"] > 0).all()


# ------------------------------------------------------------------------------------------------
# Batch_Generator / Data / preprocess.py over a miniature MSCOCO tree (utils/batch_gen.py, utils/data.py)
# ------------------------------------------------------------------------------------------------
import os
import pickle
import subprocess
import sys

import pytest

from vae_captioning_amd.utils.batch_gen import Batch_Generator, feed_dict
from vae_captioning_amd.utils.image_utils import load_image
from vae_captioning_amd.utils.parameters import Parameters

from . import coco_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def coco(tmp_path, monkeypatch):
    root = coco_fixture.build(tmp_path / "coco")
    monkeypatch.chdir(tmp_path)  # ./pickles, ./obj_vectors are relative to the working directory, as in the reference
    return root


def _caps(root, split, word2idx=None):
    c = Captions(root + "annotations/captions_%s2014.json" % split)
    if word2idx is not None:
        c.index_captions(word2idx)
    return c


def test_batch_generator_feature_path(coco):
    tr = _caps(coco, "train")
    d = Dictionary(tr.captions, 1)
    tr.index_captions(d.word2idx)
    feats = {fn: np.full((1, 4096), i, np.float32) for i, fn in enumerate(sorted(tr.captions))}
    cv = {fn: np.arange(91, dtype=np.float64) + i for i, fn in enumerate(sorted(tr.captions))}
    g = Batch_Generator(coco + "images/train2014/", coco + "annotations/captions_train2014.json", tr, 4, feature_dict=feats,
                        cluster_vectors=cv)
    items = list(g.next_batch(use_obj_vectors=True, num_captions=5))
    assert [it[0].shape for it in items] == [(4, 4096), (2, 4096)]                  # ragged last batch is yielded
    images, (ins, lab), lens, cl_v = items[0]
    assert ins.shape == lab.shape and ins.ndim == 3 and ins.shape[:2] == (4, 5) and lens.shape == (4, 5) and cl_v.shape == (4, 91)
    bos, eos = d.word2idx["<BOS>"], d.word2idx["<EOS>"]
    assert np.all(ins[:, :, 0] == bos)
    for b in range(4):
        for k in range(5):
            L = int(lens[b, k])
            assert lab[b, k, L - 1] == eos and np.all(lab[b, k, L:] == 0) and np.array_equal(ins[b, k, 1:L], lab[b, k, :L - 1])
    fd = feed_dict(images, (ins, lab), lens, cl_v, 5, False)
    assert fd["cap_dec"].shape == (20, ins.shape[2]) and fd["c_v"].shape == (20, 90) and fd["features"].shape == (4, 4096)
    assert np.array_equal(fd["c_v"][0], fd["c_v"][4]) and fd["c_v"][0, 0] == cl_v[0, 1]   # repeated per caption, column 0 dropped
    one = next(iter(g.next_batch(num_captions=1)))
    assert one[1][0].ndim == 2 and one[2].shape == (4,) and len(one[3]) == 0          # random single caption: squeezed, no c_v
    # validation / test generators
    val = _caps(coco, "val", d.word2idx)
    vfe = {fn: np.zeros((1, 4096), np.float32) for fn in val.captions}
    gv = Batch_Generator(coco + "images/val2014/", coco + "annotations/captions_val2014.json", val, 3, feature_dict=vfe, get_image_ids=True)
    out = list(gv.next_val_batch(get_image_ids=True))
    assert [len(o[3]) for o in out] == [3, 1] and sorted(sum((o[3] for o in out), [])) == sorted(val.filename_to_imid.values())
    gt = Batch_Generator(coco + "images/test2014/", train_cap_json=coco + "annotations/image_info_test2014.json", batch_size=2,
                         feature_dict={os.path.basename(p): np.zeros((1, 4096), np.float32) for p in os.listdir(coco + "images/test2014/")},
                         get_image_ids=True, get_test_ids=True)
    images, ids, cl = next(iter(gt.next_test_batch()))
    assert images.shape == (2, 4096) and len(ids) == 2 and len(cl) == 0
    # train + part of val
    g.repartiton(val, vfe, gen_val_cap=1)
    assert len(g._iterable) == 6 + 3 and len(g.unused_cap_in) == 1
    seen = sum(len(it[0]) for it in g.next_batch(num_captions=5))
    assert seen == 9


def test_image_paths_and_preprocessed_array(coco):
    tr = _caps(coco, "train")
    d = Dictionary(tr.captions, 1)
    tr.index_captions(d.word2idx)
    tdir = coco + "images/train2014/"
    g = Batch_Generator(tdir, None, tr, 3)                                            # no features, no array: decode files
    images, _, _, _ = next(iter(g.next_val_batch()))
    names = sorted(os.listdir(tdir))
    assert images.shape == (3, 224, 224, 3) and images.dtype == np.uint8
    assert np.array_equal(images[1], load_image(tdir + names[1]))
    # preprocess.py: one array + name -> row map; the generator then reads rows in increasing index order
    r = subprocess.run([sys.executable, os.path.join(ROOT, "preprocess.py"), "--coco_dir", coco.rstrip("/"), "--output_h5", "train_val.hdf5"],
                       capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout + r.stderr
    arr = np.load("train_val.npy", mmap_mode="r")
    itoi = pickle.load(open("pickles/itoi.pickle", "rb"))
    assert arr.shape == (10, 224, 224, 3) and len(itoi) == 10 and np.array_equal(arr[itoi[names[2]]], load_image(tdir + names[2]))
    g2 = Batch_Generator(tdir, None, tr, 4, use_hdf5=True, hdf5_file="train_val.hdf5")   # resolves to the .npy next to it
    assert g2.use_hdf5
    for images, _, _, _ in g2.next_batch(num_captions=5):
        assert images.dtype == np.uint8 and images.shape[1:] == (224, 224, 3)
    g3 = Batch_Generator(tdir, None, tr, 4, use_hdf5=True, hdf5_file="missing.hdf5")
    assert not g3.use_hdf5


def test_data_class_builds_vocabulary_and_generators(coco):
    from vae_captioning_amd.utils.data import Data
    p = Parameters()
    p.coco_dir, p.keep_words, p.use_hdf5 = coco, 1, False
    data = Data(p)
    assert data.num_examples == 6 and data.dictionary.word2idx["<PAD>"] == 0 and "<UNK>" in data.dictionary.word2idx
    with pytest.raises(ValueError):
        Data(p, repartiton=True)
    with pytest.raises(ValueError):
        Data(p, extract_features=True)
    gen = data.load_train_data_generator(2, fine_tune=True)
    images, (ins, lab), lens, cv = next(iter(gen.next_batch(num_captions=5)))
    assert images.shape == (2, 224, 224, 3) and ins.shape[:2] == (2, 5)
    fd = feed_dict(images, (ins, lab), lens, cv, 5, True)
    assert fd["images"].dtype == np.uint8 and fd["images"].shape == (2, 224, 224, 3) and "c_v" not in fd   # pixels stay uint8 (HDF5, preprocess.py:27-28); the device casts
    vg = data.get_valid_data(2, pretrained=False)
    assert len(list(vg.next_val_batch(get_image_ids=True))) == 2
    tg = data.get_test_data(2, pretrained=False)
    assert len(next(iter(tg.next_test_batch()))[1]) == 2


def test_cluster_vectors_from_annotations_and_detections():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import prepare_cluster_vectors as pcv
    j = dict(images=[dict(id=1, file_name="a.jpg"), dict(id=2, file_name="b.jpg"), dict(id=3, file_name="c.jpg")],
             annotations=[dict(image_id=1, category_id=18), dict(image_id=1, category_id=18), dict(image_id=1, category_id=1),
                          dict(image_id=2, category_id=90)])
    cv = pcv.cluster_vectors_from_instances(j)
    assert set(cv) == {"a.jpg", "b.jpg"} and cv["a.jpg"].shape == (91,)
    assert cv["a.jpg"][18] == 0.5 and cv["a.jpg"][1] == 0.5 and cv["a.jpg"].sum() == 1.0 and cv["b.jpg"][90] == 1.0
    ts = {"t.jpg": dict(classes=[[3.0, 7.0, 9.0]], scores=[[0.9, 0.4, 0.6]]), "u.jpg": dict(classes=[[5.0]], scores=[[0.1]])}
    tv = pcv.cluster_vectors_from_scores(ts)
    assert tv["t.jpg"][3] == 0.5 and tv["t.jpg"][9] == 0.5 and tv["t.jpg"][7] == 0 and tv["u.jpg"].sum() == 0


# ---------------------------------------------------------------------------------------------------------------------
# data-parallel sharding of the training generators (main.py --coco_dir / --captions_json under torchrun)
def test_shard_ranges_partition_every_global_batch():
    from vae_captioning_amd.utils.batch_gen import shard_ranges
    n, bs = 103, 8
    assert shard_ranges(n, bs) == [(s, min(s + bs, n)) for s in range(0, n, bs)]          # the reference's chunks, ragged last one
    for world in (2, 4):
        per_rank = [shard_ranges(n, bs, (r, world)) for r in range(world)]
        steps = n // (bs * world)
        assert all(len(p) == steps for p in per_rank)                                      # every rank steps the same number of times
        for k in range(steps):
            rows = sorted(i for p in per_rank for i in range(*p[k]))
            assert rows == list(range(k * bs * world, (k + 1) * bs * world))               # disjoint, cover the global batch exactly
            assert all(hi - lo == bs for p in per_rank for lo, hi in [p[k]])


def test_two_ranks_see_disjoint_examples_of_one_shuffle():
    caps = {"im%03d" % i: [[1, 3 + i % 5, 2]] for i in range(40)}
    feats = {k: np.full((1, 8), i, np.float32) for i, k in enumerate(caps)}
    gens = [BatchGenerator(caps, feats, 4, seed=7, shard=(r, 2)) for r in range(2)]
    a, b = (list(g.next_batch(num_captions=1)) for g in gens)
    assert len(a) == len(b) == 5
    seen = []
    for x, y in zip(a, b):
        assert not set(x["names"]) & set(y["names"])
        seen += x["names"] + y["names"]
    assert sorted(seen) == sorted(caps)                 # one epoch = every example exactly once over the two ranks
    one = [n for bt in BatchGenerator(caps, feats, 8, seed=7).next_batch(num_captions=1) for n in bt["names"]]
    assert seen == one                                  # and in the order a single rank with the global batch size would read them


def test_ranks_pad_to_the_longest_caption_of_the_global_batch():
    """Data-parallel ranks must step with the SAME T (kernel shapes, step time): each rank forms the captions of the global batch and
    keeps its rows, so rank r's arrays are exactly rows [r*bs, (r+1)*bs) of the single-process batch of size world*bs -- also with
    random_select (num_captions = 1), whose draws every rank repeats identically."""
    rng = np.random.default_rng(5)
    caps = {"im%03d" % i: [[1] + list(rng.integers(3, 50, size=14 if i == 7 else int(rng.integers(2, 6)))) + [2] for _ in range(int(rng.integers(1, 6)))] for i in range(48)}
    caps = {k: [[int(t) for t in c] for c in v] for k, v in caps.items()}
    feats = {k: np.full((1, 8), i, np.float32) for i, k in enumerate(caps)}
    for nc in (1, 5):
        ranks = [list(BatchGenerator(caps, feats, 4, seed=11, shard=(r, 3)).next_batch(num_captions=nc)) for r in range(3)]
        whole = list(BatchGenerator(caps, feats, 12, seed=11).next_batch(num_captions=nc))
        assert len(whole) == len(ranks[0]) == 4
        Ts = set()
        for k, g in enumerate(whole):
            for r in range(3):
                b = ranks[r][k]
                assert b["cap_dec"].shape[1] == g["cap_dec"].shape[1]                      # the global T on every rank
                for key in ("cap_dec", "cap_enc", "lengths"):
                    np.testing.assert_array_equal(b[key], g[key][r * 4 * nc:(r + 1) * 4 * nc], err_msg="%s nc=%d" % (key, nc))
                np.testing.assert_array_equal(b["features"], g["features"][r * 4:(r + 1) * 4])
            Ts.add(g["cap_dec"].shape[1])
        assert len(Ts) > 1   # (T does change from batch to batch in this data)


def test_imagenet_npz_is_assigned_in_sorted_key_order(tmp_path):
    """utils/image_embeddings.py:240-246 (quirk Q18): the first 30 ALPHABETICALLY sorted arrays go to `parameters` in creation
    order, whatever order the archive stores them in; fc8_* (sorted last) are skipped."""
    from vae_captioning_amd import spec
    from vae_captioning_amd.trainer import imagenet_weights
    keys = ["conv%d_%d_%s" % (b, i, s) for b, reps in ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3)) for i in range(1, reps + 1) for s in ("W", "b")]
    keys += ["fc6_W", "fc6_b", "fc7_W", "fc7_b", "fc8_W", "fc8_b"]
    rng = np.random.default_rng(3)
    stored = list(keys)
    rng.shuffle(stored)
    path = str(tmp_path / "vgg16_weights.npz")
    np.savez(path, **{k: np.full((2,), float(keys.index(k)), np.float64) for k in stored})
    got = imagenet_weights(path)
    names = [n for n, _ in spec.vgg_variables()]
    assert list(got) == names and len(got) == 30
    for i, n in enumerate(names):   # sorted(keys) == keys here: conv1_1_W, conv1_1_b, ... fc7_b, fc8_W, fc8_b
        assert got[n].dtype == np.float32 and got[n][0] == float(i), n
    assert sorted(keys) == keys


def test_hdf5_image_container_when_h5py_is_available(coco, tmp_path, monkeypatch):
    """The reference's image container (preprocess.py:25-45 writes data set "images" (N, 224, 224, 3) uint8; utils/batch_gen.py:152-162
    reads it with increasing indices).  Runs wherever h5py is installed (not in the build image: there the same array is a .npy,
    covered by test_image_paths_and_preprocessed_array)."""
    h5py = pytest.importorskip("h5py")
    import preprocess
    monkeypatch.chdir(tmp_path)
    n = preprocess.build(coco, str(tmp_path / "train_val.h5"), index_path=str(tmp_path / "pickles" / "itoi.pickle"))
    with h5py.File(str(tmp_path / "train_val.h5"), "r") as f:
        assert f["images"].shape == (n, 224, 224, 3) and f["images"].dtype == np.uint8
    from vae_captioning_amd.utils.batch_gen import open_image_array
    arr = open_image_array(str(tmp_path / "train_val.h5"))
    idx = sorted({0, n - 1, n // 2})
    assert np.asarray(arr[idx]).shape == (len(idx), 224, 224, 3)
