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
