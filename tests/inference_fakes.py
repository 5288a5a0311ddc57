"""Duck-typed stand-ins for the OBJECTS the inference driver is handed (decoder, batch generators, saver, params) -- not for any
library: `inference(params, decoder, val_gen, test_gen, image_f_inputs, saver, sess)` (ops/inference.py:4-56) only calls methods on
its arguments, so the reference's own function runs on these in the build container (tests/golden/make_ref_fixtures.py) and this
build's counterpart runs on the same objects in tests/test_ref_fixtures.py.  Every call is recorded in `trace`."""
import numpy as np

CASES = [dict(use_c_v=False, prior="Normal", sample_gen="greedy"), dict(use_c_v=False, prior="GMM", sample_gen="beam_search"),
         dict(use_c_v=True, prior="AG", sample_gen="beam_search"), dict(use_c_v=True, prior="Normal", sample_gen="greedy"),
         dict(use_c_v=False, prior="AG", sample_gen="sample")]


class Params(object):
    checkpoint, fine_tune, beam_size, gen_name = "ck7", False, 3, "fx"

    def __init__(self, use_c_v, prior, sample_gen):
        self.use_c_v, self.prior, self.sample_gen = use_c_v, prior, sample_gen


def _cv(ids):
    """[B, 91] cluster vectors: column 0 is the row marker the model never sees (ops/inference.py:17-19)"""
    a = np.zeros((len(ids), 91), np.float32)
    for r, i in enumerate(ids):
        a[r, 0] = 1000.0 + i
        a[r, 1 + i % 90] = 0.5
        a[r, 1 + (7 * i) % 90] += 0.25
    return a


class Gen(object):
    def __init__(self, trace, batches):
        self.trace, self.batches = trace, batches

    def next_val_batch(self, get_image_ids=False, use_obj_vectors=False):
        self.trace.append(["next_val_batch", bool(get_image_ids), bool(use_obj_vectors)])
        for ids in self.batches:
            yield np.full((len(ids), 4), float(ids[0]), np.float32), None, None, list(ids), _cv(ids)

    def next_test_batch(self, use_obj_vectors=False):
        self.trace.append(["next_test_batch", bool(use_obj_vectors)])
        for ids in self.batches:
            yield np.full((len(ids), 4), float(ids[0]), np.float32), list(ids), _cv(ids)


def _summ(c_v):
    a = np.asarray(c_v)
    return [list(a.shape), round(float(a.sum()), 4), round(float(a[:, 0].sum()), 4)]


class Decoder(object):
    def __init__(self, trace):
        self.trace = trace

    def beam_search(self, sess, image_ids, f_images, placeholder, c_v, beam_size=2):
        self.trace.append(["beam_search", list(image_ids), list(np.asarray(f_images).shape), _summ(c_v), int(beam_size), placeholder])
        return [{"image_id": int(i), "caption": "beam %d" % i} for i in image_ids]

    def online_inference(self, sess, image_ids, f_images, placeholder, c_v=None):
        self.trace.append(["online_inference", list(image_ids), list(np.asarray(f_images).shape), _summ(c_v), placeholder])
        return [{"image_id": int(i), "caption": "greedy %d" % i} for i in image_ids], None


class Saver(object):
    def __init__(self, trace):
        self.trace = trace

    def restore(self, sess, path):
        self.trace.append(["restore", sess, path])


def run(inference_fn, case, workdir):
    """-> {"trace": [...], "val": json of ./val_fx.json, "test": json of ./test_fx.json}; a stale val file is there to be replaced"""
    import contextlib
    import io
    import json
    import os
    trace = []
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        with open("val_fx.json", "w") as fh:
            fh.write("stale")
        with contextlib.redirect_stdout(io.StringIO()):
            inference_fn(Params(**case), Decoder(trace), Gen(trace, [[3, 5, 8], [13]]), Gen(trace, [[21, 34], [55, 89, 144]]), "PH", Saver(trace), "SESS")
        return dict(trace=trace, val=json.load(open("val_fx.json")), test=json.load(open("test_fx.json")))
    finally:
        os.chdir(cwd)
