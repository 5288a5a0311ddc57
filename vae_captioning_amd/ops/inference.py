"""`inference` of the reference's ops/inference.py:4-56: restore the checkpoint, caption the validation
and test sets, write ./val_{gen_name}.json and ./test_{gen_name}.json ([{"image_id":…, "caption":…}])."""
import json
import os

import numpy as np


def _drop_col0(c_v):
    c_v = np.asarray(c_v)
    return c_v[:, 1:] if c_v.ndim == 2 and c_v.shape[1] else c_v


def inference(params, decoder, val_gen, test_gen, image_f_inputs=None, saver=None, sess=None):
    """`saver` is anything with .restore(sess, path) (a Trainer works: restore(path)); None = weights already loaded."""
    if saver is not None:
        print("Restoring from checkpoint")
        path = "./checkpoints/{}.ckpt".format(params.checkpoint)
        try:
            saver.restore(sess, path)
        except TypeError:
            saver.restore(path)
    if not params.fine_tune:
        print("Using prepared features for generation. If you want to use fine-tuned VGG16 feature extractor, "
              "need to specify --fine_tune parameter.")
    captions_gen = []
    print("Generating captions for val file")
    for images, _, _, image_ids, c_v in val_gen.next_val_batch(get_image_ids=True, use_obj_vectors=params.use_c_v):
        if params.use_c_v or params.prior in ("GMM", "AG"):
            c_v = _drop_col0(c_v)  # 0 element doesnt matter (ops/inference.py:17-19)
        if params.sample_gen == "beam_search":
            sent = decoder.beam_search(sess, image_ids, images, image_f_inputs, c_v, beam_size=params.beam_size)
        else:
            sent, _ = decoder.online_inference(sess, image_ids, images, image_f_inputs, c_v=c_v)
        captions_gen += sent
    print("Generated {} captions".format(len(captions_gen)))
    val_gen_file = "./val_{}.json".format(params.gen_name)
    if os.path.exists(val_gen_file):
        os.remove(val_gen_file)
    with open(val_gen_file, "w") as wj:
        print("saving val json file into ", val_gen_file)
        json.dump(captions_gen, wj)
    if test_gen is None:
        return
    captions_gen = []
    print("Generating captions for test file")
    for images, image_ids, c_v in test_gen.next_test_batch(params.use_c_v):
        if params.use_c_v:
            c_v = _drop_col0(c_v)
        sent, _ = decoder.online_inference(sess, image_ids, images, image_f_inputs, c_v=c_v)
        captions_gen += sent
    test_gen_file = "./test_{}.json".format(params.gen_name)
    if os.path.exists(test_gen_file):
        os.remove(test_gen_file)
    with open(test_gen_file, "w") as wj:
        print("saving test json file into", test_gen_file)
        json.dump(captions_gen, wj)
