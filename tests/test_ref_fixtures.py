"""Host-side modules and the oracle's TopN / Beam against fixtures computed by THE REFERENCE's own modules
(tests/golden/make_ref_fixtures.py imports /root/reference/utils/{captions,caption_utils,top_n,parameters}.py in the build
container and stores inputs + outputs as json).  These pin, against the reference itself rather than hand-worked answers:
tokenisation and vocabulary order (utils/captions.py:38-126), preprocess_captions (utils/caption_utils.py:4-25),
TopN / Beam heap behaviour under exact score ties (utils/top_n.py:4-72), every Parameters flag cast
(utils/parameters.py:75-164) and the inference driver's calls into the decoders (ops/inference.py:4-56)."""
import json
import os

import numpy as np
import pytest

from oracle import decode as OD
from vae_captioning_amd.utils import batch_gen, captions as C, top_n as T
from vae_captioning_amd.utils.parameters import Parameters

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
load = lambda name: json.load(open(os.path.join(G, name)))


# ------------------------------------------------------------------ captions / vocabulary
def test_tokeniser_matches_reference_on_every_caption():
    fx = load("ref_captions.json")
    caps = C.Captions(fx["coco"], 100)
    assert {k: v for k, v in caps.captions.items()} == fx["tokens"]
    assert list(caps.captions.keys()) == [k for k in dict.fromkeys(
        {i["id"]: i["file_name"] for i in fx["coco"]["images"]}[a["image_id"]] for a in fx["coco"]["annotations"])]
    assert caps.filename_to_imid == fx["filename_to_imid"]
    assert caps.num_captions == fx["num_captions"]
    for ann in fx["coco"]["annotations"]:  # the free function is what gen_caption.py uses
        assert C.tokenize(ann["caption"])[0] == "<BOS>" and C.tokenize(ann["caption"])[-1] == "<EOS>"


@pytest.mark.parametrize("keep", [1, 2, 3])
def test_vocabulary_ids_match_reference(keep):
    fx = load("ref_captions.json")
    ref = fx["vocab"][str(keep)]
    d = C.Dictionary(fx["tokens"], keep)
    assert d.word2idx == ref["word2idx"]            # (-count, word) order, ids from 1, <PAD> = 0, <UNK> kept regardless of count
    assert d.idx2word == {i: w for w, i in ref["word2idx"].items()}
    assert d.vocab_size == ref["vocab_size"] and len(d) == ref["len"]
    assert d.seq2dx(["<BOS>", "<EOS>"]) == ref["seq2dx_bos_eos"]


def test_index_captions_matches_reference_including_unk():
    fx = load("ref_captions.json")
    caps = C.Captions(fx["coco"], 100)
    d = C.Dictionary(caps.captions, 3)
    got = caps.index_captions(d.word2idx)
    assert {k: v for k, v in got.items()} == fx["indexed_keep3"]
    unk = d.word2idx["<UNK>"]
    assert any(unk in cap for caps_ in got.values() for cap in caps_)


# ------------------------------------------------------------------ preprocess_captions
def test_preprocess_captions_matches_reference():
    for case in load("ref_preprocess_captions.json"):
        ins, lab, lens = (np.array(case[k], np.int32) for k in ("inputs", "labels", "lengths"))
        cv = np.array(case["cv"], np.float64)
        out = batch_gen.preprocess_captions(ins, lab, lens, cv if cv.size else None)
        np.testing.assert_array_equal(out["cap_dec"], np.array(case["out_inputs"]))
        np.testing.assert_array_equal(out["cap_enc"], np.array(case["out_labels"]))
        np.testing.assert_array_equal(out["lengths"], np.array(case["out_lengths"]))
        if cv.size:
            np.testing.assert_allclose(out["c_v"], np.array(case["out_cv"]), rtol=1e-7)   # (this build stores float32)
        else:
            assert "c_v" not in out and case["out_cv"] == []


# ------------------------------------------------------------------ TopN / Beam
@pytest.mark.parametrize("impl", [T, OD], ids=["product", "oracle"])
def test_topn_scripts_match_reference(impl):
    fx = load("ref_topn.json")
    for sc in fx["scripts"]:
        t = impl.TopN(sc["n"])
        sizes = []
        for k, s in enumerate(sc["scores"]):
            t.push(impl.Beam([k], None, s, s))
            sizes.append(t.size())
        assert sizes == sc["sizes"]
        assert [b.sentence[0] for b in t.extract(sort=sc["sort"])] == sc["kept"], sc
    t = impl.TopN(3)
    ident = 0
    for rnd in fx["rounds"]:
        for s in rnd["scores"]:
            t.push(impl.Beam([ident], None, s, s))
            ident += 1
        assert [b.sentence[0] for b in t.extract()] == rnd["kept"]
        t.reset()


def replay_beam_rounds(impl, rec):
    """The bookkeeping of vae_model/decoder.py:254-293 on the fixture's top-k tables; yields (partial, complete) heaps per round."""
    n, B, eos, bos, lnf = rec["n"], rec["B"], rec["eos"], rec["bos"], rec["len_norm_f"]
    partial = [impl.TopN(n) for _ in range(B)]
    complete = [impl.TopN(n) for _ in range(B)]
    for b in range(B):
        partial[b].push(impl.Beam([bos], 0, 0.0, 0.0))
    for rnd in rec["rounds"]:
        tv, ti = np.array(rnd["top_p"], np.float32), np.array(rnd["top_i"], np.int32)
        for b in range(B):
            lst = partial[b].extract()
            partial[b].reset()
            for i, bm in enumerate(lst):
                for w, pw in zip(ti[b * n + i], tv[b * n + i]):
                    if pw < 1e-12:
                        continue
                    s = bm.sentence + [int(w)]
                    lp = float(bm.logprob) + float(np.log(np.float32(pw)))
                    (complete if w == eos else partial)[b].push(impl.Beam(s, i, lp, lp / len(s) ** lnf if w == eos else lp))
        yield rnd, partial, complete


def heap_items(t):
    return t._heap if hasattr(t, "_heap") else t._data


@pytest.mark.parametrize("impl", [T, OD], ids=["product", "oracle"])
def test_beam_rounds_match_reference_heaps(impl):
    for rec in load("ref_beam_rounds.json"):
        for rnd, partial, complete in replay_beam_rounds(impl, rec):
            for name, heaps in (("partial", partial), ("complete", complete)):
                for b, h in enumerate(heaps):
                    got = [dict(sentence=bm.sentence, parent=int(bm.state), logprob=bm.logprob, score=bm.score) for bm in heap_items(h)]
                    assert got == rnd[name][b], (rec["n"], name, b)


# ------------------------------------------------------------------ Parameters
# attributes whose values legitimately differ: none of the reference's; this build only ADDS attributes
def test_parameter_defaults_match_reference():
    fx = load("ref_parameters.json")
    p = Parameters()
    for k, v in fx["defaults"].items():
        assert getattr(p, k) == v, k


def test_parse_args_casts_match_reference_for_every_flag():
    fx = load("ref_parameters.json")
    seen = set()
    for case in fx["cases"]:
        env = dict(os.environ)
        try:
            p = Parameters().parse_args(case["argv"])
            for k, v in case["attrs"].items():
                got = getattr(p, k)
                assert got == v and (isinstance(got, bool) == isinstance(v, bool) or k == "save_params"), (case["argv"], k, got, v)
            assert os.environ.get("HIP_VISIBLE_DEVICES") == case["cuda_visible_devices"]  # Q19: --gpu -> device mask
        finally:
            os.environ.clear()
            os.environ.update(env)
        seen.update(a for a in case["argv"] if a.startswith("--"))
    assert len(seen) == 26, sorted(seen)  # every flag of utils/parameters.py:75-132 is exercised


# ------------------------------------------------------------------ the caller of the decode path
@pytest.mark.parametrize("k", range(5))
def test_inference_driver_makes_the_reference_drivers_calls(k, tmp_path):
    """ops/inference.py:4-56 run BY THE REFERENCE on recording stand-ins for its arguments (tests/inference_fakes.py) against this
    build's `vae_captioning_amd.ops.inference.inference` on the same objects: checkpoint path, generator keywords, decoder method
    per image set (beam search only for the validation set), the cluster-vector columns that reach the decoder (column 0 is cut
    for c_v / GMM / AG models on the validation set, for c_v models only on the test set -- ops/inference.py:17-19,43-44), and both
    result files (a stale one is replaced)."""
    from tests import inference_fakes as F
    from vae_captioning_amd.ops.inference import inference
    fx = load("ref_inference.json")[k]
    assert fx["case"] == F.CASES[k]
    got = F.run(inference, fx["case"], str(tmp_path))
    assert got["trace"] == fx["result"]["trace"]
    assert got["val"] == fx["result"]["val"] and got["test"] == fx["result"]["test"]
