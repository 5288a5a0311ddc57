"""Caption the validation and test image sets and store them as COCO-style result files.

Counterpart of the reference's `inference(params, decoder, val_gen, test_gen, image_f_inputs, saver, sess)`
(ops/inference.py:4-56): same argument list, same output files -- `./val_{gen_name}.json` and `./test_{gen_name}.json`,
each a list of `{"image_id": ..., "caption": ...}` -- produced by the batched on-device decoders of
`vae_model/decoder.py`.  Validation images use `params.sample_gen` (beam search or greedy / sampling); the test set is
always decoded with `online_inference`, as in the reference."""
import json
import os

import numpy as np


def _cluster_rows(c_v, wanted):
    """Batch generator rows are 91-vectors; the model takes columns 1..90 (ops/inference.py:17-19, main.py:236)."""
    if not wanted:
        return c_v
    a = np.asarray(c_v)
    return a[:, 1:] if a.ndim == 2 and a.shape[1] > 0 else a


def _decode(decoder, params, sess, placeholder, ids, images, c_v, allow_beam):
    if allow_beam and params.sample_gen == "beam_search":
        return decoder.beam_search(sess, ids, images, placeholder, c_v, beam_size=params.beam_size)
    return decoder.online_inference(sess, ids, images, placeholder, c_v=c_v)[0]


def _store(path, records):
    if os.path.exists(path):
        os.remove(path)
    with open(path, "w") as fh:
        json.dump(records, fh)
    print("wrote %d captions to %s" % (len(records), path))


def _restore(saver, sess, prefix):
    """`saver` may be a tf.train.Saver look-alike (restore(sess, path)) or a Trainer (restore(path))."""
    if saver is None:
        return
    print("Restoring from checkpoint", prefix)
    try:
        saver.restore(sess, prefix)
    except TypeError:
        saver.restore(prefix)


def inference(params, decoder, val_gen, test_gen, image_f_inputs=None, saver=None, sess=None):
    _restore(saver, sess, "./checkpoints/{}.ckpt".format(params.checkpoint))
    if not params.fine_tune:
        print("Captioning from precomputed fc2 features; pass --fine_tune to run the fine-tuned VGG16 on the images.")
    val_cv = params.use_c_v or params.prior in ("GMM", "AG")
    records = []
    for images, _caps, _lens, ids, c_v in val_gen.next_val_batch(get_image_ids=True, use_obj_vectors=params.use_c_v):
        records += _decode(decoder, params, sess, image_f_inputs, ids, images, _cluster_rows(c_v, val_cv), allow_beam=True)
    _store("./val_{}.json".format(params.gen_name), records)
    if test_gen is None:
        return
    records = []
    for images, ids, c_v in test_gen.next_test_batch(params.use_c_v):
        records += _decode(decoder, params, sess, image_f_inputs, ids, images, _cluster_rows(c_v, params.use_c_v), allow_beam=False)
    _store("./test_{}.json".format(params.gen_name), records)
