"""-m gpu: batched on-device greedy decoding and beam search vs the oracle's per-image
restatement of vae_model/decoder.py:145-320.  Token ids must be IDENTICAL (integer work);
beam scores agree to 1e-4 (fp32 probabilities)."""
import numpy as np
import pytest

from oracle import decode as od
from vae_captioning_amd import spec
from vae_captioning_amd.engine import CaptionEngine
from vae_captioning_amd.generate import CaptionGenerator
from vae_captioning_amd.utils.parameters import Parameters

pytestmark = pytest.mark.gpu
BOS, EOS = 1, 2


def setup(lib, seed, **kw):
    p = Parameters()
    p.embed_size, p.encoder_hidden, p.decoder_hidden = 32, 64, 64
    p.latent_size, p.gen_z_samples, p.cnn_feature_size = 10, 4, 48
    p.mode, p.num_captions = "inference", 1
    for k, v in kw.items():
        setattr(p, k, v)
    V, B = 40, 6
    rng = np.random.default_rng(seed)
    P0 = spec.init_caption_params(p, V, seed=seed)
    for k in P0:  # larger weights -> peaked distributions, <EOS> reachable
        P0[k] = (P0[k] * 3).astype(np.float32) if not k.endswith("bias") else rng.normal(0, 0.5, P0[k].shape).astype(np.float32)
    feats = np.maximum(rng.standard_normal((B, p.cnn_feature_size)), 0).astype(np.float32)
    cv = np.zeros((B, 90), np.float32)
    for b in range(B - 1):  # last image: empty cluster vector (AG fallback branch, Q16)
        cv[b, rng.choice(90, size=2, replace=False)] = 0.5
    eps = rng.standard_normal((p.gen_z_samples, B, p.latent_size)).astype(np.float32)
    eng = CaptionEngine(p, V, lib=lib)
    eng.load_params(P0)
    P64 = {k: v.astype(np.float64) for k, v in P0.items()}
    cm = od.init_clusters(90, p.latent_size).astype(np.float64) if p.prior == "AG" else None
    return p, eng, CaptionGenerator(eng), P64, feats, cv, eps, cm


CASES = [dict(no_encoder=True), dict(prior="Normal"), dict(prior="AG", use_c_v=True), dict(prior="GMM")]


@pytest.mark.parametrize("kw", CASES, ids=lambda k: "-".join("%s=%s" % i for i in k.items()))
def test_greedy_token_ids_identical(lib, kw):
    p, eng, gen, P64, feats, cv, eps, cm = setup(lib, 5, **kw)
    got = gen.greedy(feats, cv if spec.uses_ci(p) else None, eps, BOS, EOS, max_len=12)
    for b in range(feats.shape[0]):
        ref = od.greedy(P64, p, feats[b].astype(np.float64), cv[b].astype(np.float64), eps[:, b:b + 1].astype(np.float64), BOS, EOS,
                        c_means=cm, max_len=12)
        assert got[b] == ref, (b, got[b], ref)


@pytest.mark.parametrize("kw", CASES, ids=lambda k: "-".join("%s=%s" % i for i in k.items()))
@pytest.mark.parametrize("beam", [2, 5])
def test_beam_search_matches_oracle(lib, kw, beam):
    p, eng, gen, P64, feats, cv, eps, cm = setup(lib, 7, **kw)
    got = gen.beam_search(feats, cv if spec.uses_ci(p) else None, eps, BOS, EOS, beam_size=beam, max_len=10)
    for b in range(feats.shape[0]):
        sents, scores = od.beam_search(P64, p, feats[b].astype(np.float64), cv[b].astype(np.float64), eps[:, b:b + 1].astype(np.float64),
                                       BOS, EOS, c_means=cm, beam_size=beam, max_len=10)
        assert [s for s, _ in got[b]] == sents, (b, got[b], sents, scores)
        np.testing.assert_allclose([sc for _, sc in got[b]], scores, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("k,cols,ld", [(9, 300, 300), (5, 300, 300), (8, 1003, 1004), (5, 10000, 10000), (3, 257, 259), (1, 40, 40), (5, 6, 8)],
                         ids=["k9-k-pass-kernel", "k5", "k8-ragged-vector-tail", "k5-vocab", "k3-unaligned-pitch", "k1", "k5-of-6"])
def test_topk_is_a_stable_descending_sort_prefix(lib, k, cols, ld):
    """(value descending, index ascending) = the first k entries of a STABLE sort on -p (decoder.py:273-276), with many ties;
    k <= 8 runs the single-pass kernel (16-byte loads when the pitch allows, per-thread sorted lists merged through LDS)."""
    import torch
    from .gpu_util import P, dev, host, stream
    rng = np.random.default_rng(k * 1000 + cols)
    R = 7
    x = np.full((R, ld), 99.0, np.float32)  # (the padding columns hold a LARGER value: they must never be selected)
    x[:, :cols] = rng.integers(0, 6, size=(R, cols)).astype(np.float32)
    x[0, :cols] = 2.0                       # a constant row: the answer is indices 0 .. k-1
    tv = torch.empty((R, k), dtype=torch.float32, device="cuda")
    ti = torch.empty((R, k), dtype=torch.int32, device="cuda")
    lib.vc_topk_rows_f32(stream(), P(dev(x)), R, cols, ld, k, P(tv), P(ti))
    for r in range(R):
        ref = sorted(enumerate(x[r, :cols]), key=lambda t: -t[1])[:k]  # Python's sort is stable
        assert host(ti)[r].tolist() == [i for i, _ in ref]
        assert host(tv)[r].tolist() == [float(v) for _, v in ref]


@pytest.mark.parametrize("k,V,ld", [(5, 10000, 10000), (8, 1000, 1004), (3, 257, 259), (1, 40, 40), (5, 6, 8), (5, 11313, 11316), (2, 12292, 12292)],
                         ids=["k5-vocab", "k8-padded-pitch", "k3-unaligned-three-pass", "k1", "k5-of-6", "observed-vocab-three-pass", "beyond-the-register-form"])
def test_fused_softmax_topk_equals_softmax_then_topk_bit_for_bit(lib, k, V, ld):
    """round 6: one beam-search round reads its logits once (vc_softmax_topk_rows_f32) instead of writing [rows, V] probabilities
    and reading them back (vae_model/decoder.py:248-276).  Rows with many EQUAL logits (ties -> lower index first), a constant row,
    a row with one dominant word (p == 1.0 exactly, everything else underflows to equal zeros)."""
    import torch
    from .gpu_util import P, dev, host, stream
    rng = np.random.default_rng(k * 7 + V)
    R = 9
    x = np.full((R, ld), 50.0, np.float32)   # (padding columns hold a LARGER value: they must never be read)
    x[:, :V] = rng.standard_normal((R, V)).astype(np.float32) * 3
    x[1, :V] = np.round(x[1, :V])            # ties
    x[2, :V] = 0.25                          # constant row
    x[3, :V] = -200.0
    x[3, V // 2] = 100.0                     # one word takes everything
    if V >= 1000:   # two-level rows: 300 / 200 words share the top probability (more / fewer than the selection's candidate list holds)
        for r, cnt in ((6, 300), (7, 200)):
            x[r, :V] = 0.0
            x[r, rng.choice(V, size=cnt, replace=False)] = 1.0
    dx = dev(x)
    probs = torch.empty((R, ld), dtype=torch.float32, device="cuda")
    tv, ti = [torch.empty((R, k), dtype=torch.float32, device="cuda") for _ in range(2)], [torch.empty((R, k), dtype=torch.int32, device="cuda") for _ in range(2)]
    lib.vc_softmax_rows_f32(stream(), P(dx), R, V, ld, P(probs), ld)
    lib.vc_topk_rows_f32(stream(), P(probs), R, V, ld, k, P(tv[0]), P(ti[0]))
    lib.vc_softmax_topk_rows_f32(stream(), P(dx), R, V, ld, k, P(tv[1]), P(ti[1]))
    assert np.array_equal(host(ti[0]), host(ti[1])) and np.array_equal(host(tv[0]).view(np.uint32), host(tv[1]).view(np.uint32))
    # and against the definition, in fp64 (values to 1e-6, the index set where the probabilities are distinct)
    ref = np.exp(x[:, :V].astype(np.float64) - x[:, :V].max(1, keepdims=True))
    ref /= ref.sum(1, keepdims=True)
    for r in (0, 4, 5):
        order = sorted(range(V), key=lambda i: -ref[r, i])[:k]
        assert host(ti[1])[r].tolist() == order
        np.testing.assert_allclose(host(tv[1])[r], ref[r, order], rtol=2e-6)
    assert host(ti[1])[2].tolist() == list(range(k)) and host(ti[1])[3, 0] == V // 2 and host(tv[1])[3, 0] == 1.0
    if V >= 1000:
        for r in (6, 7):
            assert host(ti[1])[r].tolist() == np.nonzero(x[r, :V] == 1.0)[0][:k].tolist()


@pytest.mark.parametrize("kw", CASES, ids=lambda k: "-".join("%s=%s" % i for i in k.items()))
def test_replayed_calls_decode_the_inputs_of_the_call_not_of_the_capture(lib, kw):
    """From the second call of a shape on, `init_state` and the rounds replay hipGraphs over persistent input buffers (features, cluster
    vectors, the AG prior means computed on the host, injected noise).  A generator that has captured its graphs on one batch must
    decode ANOTHER batch of the same shape exactly as a fresh generator does -- greedy and beam search, every prior."""
    p, eng, gen, P64, feats, cv, eps, cm = setup(lib, 21, **kw)
    rng = np.random.default_rng(99)
    c = cv if spec.uses_ci(p) else None
    run = lambda g, f, c_, e_: (g.greedy(f, c_, e_, BOS, EOS, max_len=11), g.beam_search(f, c_, e_, BOS, EOS, beam_size=3, max_len=11))
    first = run(gen, feats, c, eps)
    assert run(gen, feats, c, eps) == first                      # (replay on the same inputs)
    feats2 = np.maximum(rng.standard_normal(feats.shape), 0).astype(np.float32)
    eps2 = rng.standard_normal(eps.shape).astype(np.float32)
    cv2 = np.zeros_like(cv)
    for b in range(cv.shape[0]):
        cv2[b, rng.choice(90, size=3, replace=False)] = 0.3
    c2 = cv2 if spec.uses_ci(p) else None
    second = run(gen, feats2, c2, eps2)                           # replayed graphs, new inputs
    assert second == run(CaptionGenerator(eng), feats2, c2, eps2) and second != first
    assert run(gen, feats, c, eps) == first


@pytest.mark.parametrize("kw", [dict(prior="GMM"), dict(no_encoder=True)], ids=["gmm", "no-encoder"])
def test_captured_rounds_generate_what_the_eager_loop_generates(lib, kw, monkeypatch):
    """greedy and beam search replay hipGraphs of four decoder rounds; VC_DECODE_GRAPH=0 runs the same launches one by one.  Same
    token ids, same beams, same scores (bitwise), also when the generator is called twice (cached greedy graph) and with a batch
    size change in between."""
    p, eng, gen, P64, feats, cv, eps, cm = setup(lib, 9, **kw)
    c = cv if spec.uses_ci(p) else None
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("VC_DECODE_GRAPH", mode)
        g = CaptionGenerator(eng)
        out[mode] = (g.greedy(feats, c, eps, BOS, EOS, max_len=14), g.beam_search(feats, c, eps, BOS, EOS, beam_size=3, max_len=14),
                     g.greedy(feats[:4], c[:4] if c is not None else None, eps[:, :4], BOS, EOS, max_len=14),
                     g.greedy(feats, c, eps, BOS, EOS, max_len=14), g.beam_search(feats, c, eps, BOS, EOS, beam_size=3, max_len=14, check_every=0))
    assert out["1"] == out["0"]
    assert out["1"][0] == out["1"][3] and out["1"][1] == out["1"][4] and out["1"][2] == out["1"][0][:4]


def test_multinomial_inverse_cdf_matches_numpy(lib):
    import torch
    from .gpu_util import P, dev, host, stream
    rng = np.random.default_rng(3)
    R, V, temp = 64, 1003, 0.7
    logits = (rng.standard_normal((R, V)) * 2).astype(np.float32)
    u = rng.random(R).astype(np.float32)
    u[:3] = [0.0, 0.999999, 0.5]
    out = torch.zeros(R, dtype=torch.int32, device="cuda")
    lib.vc_multinomial_rows_f32(stream(), P(dev(logits)), R, V, V, temp, P(dev(u)), P(out))
    x = logits.astype(np.float64) / temp
    pr = np.exp(x - x.max(1, keepdims=True))
    cdf = np.cumsum(pr, axis=1)
    ref = np.array([min(V - 1, int(np.searchsorted(cdf[r], u[r] * cdf[r, -1], side="right"))) for r in range(R)])
    got = host(out)
    # fp32 vs fp64 cumulative sums may differ at a boundary: allow the neighbouring index where u sits on an edge
    def on_edge(r):  # u * total sits within fp32 round-off of a CDF step: either neighbour is right
        lo = min(int(got[r]), int(ref[r]))
        return abs(int(got[r]) - int(ref[r])) == 1 and abs(cdf[r, lo] - u[r] * cdf[r, -1]) < 1e-4 * cdf[r, -1]
    bad = [r for r in range(R) if got[r] != ref[r] and not on_edge(r)]
    assert not bad, (bad, got[bad], ref[bad])
    # distribution check: many draws of one row reproduce its softmax
    n = 200000
    lg = np.tile(logits[:1, :16], (n, 1)).copy()
    uu = torch.empty(n, device="cuda")
    lib.vc_philox_uniform_f32(stream(), P(uu), n, 5, 0, None)
    assert 0.0 <= float(uu.min()) and float(uu.max()) < 1.0
    o2 = torch.zeros(n, dtype=torch.int32, device="cuda")
    lib.vc_multinomial_rows_f32(stream(), P(dev(lg)), n, 16, 16, 1.0, P(uu), P(o2))
    freq = np.bincount(host(o2), minlength=16) / n
    p16 = np.exp(lg[0] - lg[0].max()); p16 /= p16.sum()
    assert np.abs(freq - p16).max() < 5e-3


def test_sample_mode_generates_valid_tokens(lib):
    p, eng, gen, P64, feats, cv, eps, cm = setup(lib, 9, prior="Normal")
    p.temperature = 0.8
    u = np.random.default_rng(0).random((6, feats.shape[0])).astype(np.float32)
    a = gen.sample(feats, None, eps, BOS, EOS, max_len=6, uniforms=u)
    b = gen.sample(feats, None, eps, BOS, EOS, max_len=6, uniforms=u)
    assert a == b and all(0 <= t < 40 for s in a for t in s) and all(1 <= len(s) <= 6 for s in a)


@pytest.mark.parametrize("n,rounds,L", [(1, 6, 12), (3, 6, 12), (5, 6, 12), (9, 6, 12), (16, 5, 12), (2, 70, 80)],
                         ids=["beam1", "beam3", "beam5", "beam9-two-candidate-blocks", "beam16-four-candidate-blocks", "captions-longer-than-a-wave"])
def test_beam_update_replays_heapq_including_ties(lib, n, rounds, L):
    """vc_beam_update vs the Python TopN / heapq bookkeeping of decoder.py:254-293 on synthetic top-k tables whose
    probabilities are quantised to a few values (many exact score ties, many <EOS> hits): heap ARRAY order, scores,
    captions and the parent / token rows for the next step must be identical after every round.  The kernel works one wave per image
    on blocks of 64 candidates (beam x beam of them) and copies tokens 64 at a time: beams 9 / 16 and captions of > 64 tokens cross
    those limits.  The starting state comes from vc_beam_init (on buffers filled with garbage)."""
    import torch
    from vae_captioning_amd.utils.top_n import Beam, TopN
    from .gpu_util import P, dev, host, stream
    rng = np.random.default_rng(n)
    B, eos, bos, lnf, H = 7, 2, 1, 0.7, 8
    M = B * n
    i32 = dict(dtype=torch.int32, device="cuda")
    f64 = dict(dtype=torch.float64, device="cuda")
    junk_i = lambda *shape: torch.full(shape, -7, **i32)
    junk_d = lambda *shape: torch.full(shape, 3.5, **f64)
    pcount, ccount = junk_i(B), junk_i(B)
    p_score, p_logprob, p_len = junk_d(M), junk_d(M), junk_i(M)
    sent = [junk_i(M, L), junk_i(M, L)]
    c_score, c_logprob, c_len, c_slot = junk_d(M), junk_d(M), junk_i(M), junk_i(M)
    c_free = junk_i(B)
    c_sent = junk_i(B * (n + 1), L)
    parent, tok = junk_i(M), junk_i(M)
    c_in, h_in = torch.randn(B, H, device="cuda"), torch.randn(B, H, device="cuda")
    c_out, h_out = torch.zeros(M, H, device="cuda"), torch.zeros(M, H, device="cuda")
    lib.vc_beam_init(stream(), B, n, L, bos, H, P(c_in), P(h_in), P(c_out), P(h_out), P(pcount), P(ccount), P(p_score), P(p_logprob), P(p_len),
                     P(sent[0]), P(sent[1]), P(c_score), P(c_logprob), P(c_len), P(c_slot), P(c_free), P(c_sent), P(parent), P(tok))
    assert torch.equal(c_out, c_in.repeat_interleave(n, 0)) and torch.equal(h_out, h_in.repeat_interleave(n, 0))
    assert bool((pcount == 1).all()) and bool((ccount == 0).all()) and bool((c_free == (1 << (n + 1)) - 1).all())
    assert bool((p_score == 0).all()) and bool((p_logprob == 0).all()) and bool((p_len == 1).all()) and bool((sent[0] == bos).all())
    assert all(bool((t == 0).all()) for t in (sent[1], c_score, c_logprob, c_len, c_slot, c_sent))
    assert torch.equal(parent, torch.arange(M, **i32)) and bool((tok == bos).all())
    partial = [TopN(n) for _ in range(B)]
    complete = [TopN(n) for _ in range(B)]
    for b in range(B):
        partial[b].push(Beam([bos], 0, 0.0, 0.0))
    levels = np.array([0.5, 0.25, 0.25, 0.125, 1e-13], np.float32)
    for it in range(rounds):
        tv = np.sort(rng.choice(levels, size=(M, n)).astype(np.float32), axis=1)[:, ::-1].copy()
        ti = rng.integers(2, 6, size=(M, n)).astype(np.int32)  # small vocabulary: <EOS> (= 2) comes up often
        lib.vc_beam_update(stream(), B, n, L, eos, lnf, P(dev(tv)), P(dev(ti)), P(pcount), P(ccount), P(p_score), P(p_logprob), P(p_len),
                           P(sent[it & 1]), P(sent[1 - (it & 1)]), P(c_score), P(c_logprob), P(c_len), P(c_slot), P(c_free), P(c_sent),
                           P(parent), P(tok))
        for b in range(B):  # the reference loop
            lst = partial[b].extract()
            partial[b].reset()
            for i, bm in enumerate(lst):
                for w, pw in zip(ti[b * n + i], tv[b * n + i]):
                    if pw < 1e-12:
                        continue
                    s = bm.sentence + [int(w)]
                    lp = bm.logprob + float(np.log(np.float32(pw)))  # decoder.py:282: float32 log, float64 sum
                    if w == eos:
                        complete[b].push(Beam(s, i, lp, lp / len(s) ** lnf))
                    else:
                        partial[b].push(Beam(s, i, lp, lp))
        pc, cc = host(pcount), host(ccount)
        ps, pl, sn = host(p_score).reshape(B, n), host(p_len).reshape(B, n), host(sent[1 - (it & 1)]).reshape(B, n, L)
        cs, cl, csl, cst = host(c_score).reshape(B, n), host(c_len).reshape(B, n), host(c_slot).reshape(B, n), host(c_sent).reshape(B, n + 1, L)
        par, tk = host(parent).reshape(B, n), host(tok).reshape(B, n)
        for b in range(B):
            heap = partial[b]._heap
            assert pc[b] == len(heap)
            for j, bm in enumerate(heap):  # heap ARRAY order, not sorted order
                assert sn[b, j, :pl[b, j]].tolist() == bm.sentence and ps[b, j] == bm.score
                assert par[b, j] == b * n + bm.state and tk[b, j] == bm.sentence[-1]
            cheap = complete[b]._heap
            assert cc[b] == len(cheap)
            for j, bm in enumerate(cheap):
                assert cst[b, csl[b, j], :cl[b, j]].tolist() == bm.sentence and cs[b, j] == bm.score
    assert sum(len(c._heap) for c in complete) > 0


def test_beam_update_matches_reference_topn_fixture(lib):
    """vc_beam_update against heap states recorded from THE REFERENCE's TopN / Beam classes (tests/golden/ref_beam_rounds.json,
    written by tests/golden/make_ref_fixtures.py from /root/reference/utils/top_n.py): same top-k tables in, and after every
    round the live / complete heaps must hold the same captions in the same heap-array slots with the same scores
    (score ties included), the same parent rows and next tokens.  Bit-exact: scores are float64 sums of float32 logs."""
    import json
    import os
    import torch
    from .gpu_util import P, dev, host, stream
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_beam_rounds.json")))
    for rec in fx:
        n, B, eos, bos, lnf, L = rec["n"], rec["B"], rec["eos"], rec["bos"], rec["len_norm_f"], 12
        M = B * n
        i32 = dict(dtype=torch.int32, device="cuda")
        f64 = dict(dtype=torch.float64, device="cuda")
        pcount, ccount = torch.ones(B, **i32), torch.zeros(B, **i32)
        p_score, p_logprob, p_len = torch.zeros(M, **f64), torch.zeros(M, **f64), torch.ones(M, **i32)
        sent = [torch.full((M, L), bos, **i32), torch.zeros((M, L), **i32)]
        c_score, c_logprob, c_len, c_slot = torch.zeros(M, **f64), torch.zeros(M, **f64), torch.zeros(M, **i32), torch.zeros(M, **i32)
        c_free = torch.full((B,), (1 << (n + 1)) - 1, **i32)
        c_sent = torch.zeros((B * (n + 1), L), **i32)
        parent, tok = torch.zeros(M, **i32), torch.zeros(M, **i32)
        for it, rnd in enumerate(rec["rounds"]):
            tv, ti = np.array(rnd["top_p"], np.float32), np.array(rnd["top_i"], np.int32)
            lib.vc_beam_update(stream(), B, n, L, eos, lnf, P(dev(tv)), P(dev(ti)), P(pcount), P(ccount), P(p_score), P(p_logprob), P(p_len),
                               P(sent[it & 1]), P(sent[1 - (it & 1)]), P(c_score), P(c_logprob), P(c_len), P(c_slot), P(c_free), P(c_sent),
                               P(parent), P(tok))
            pc, cc = host(pcount), host(ccount)
            ps, plp, pl = host(p_score).reshape(B, n), host(p_logprob).reshape(B, n), host(p_len).reshape(B, n)
            sn = host(sent[1 - (it & 1)]).reshape(B, n, L)
            cs, clp, cl = host(c_score).reshape(B, n), host(c_logprob).reshape(B, n), host(c_len).reshape(B, n)
            csl, cst = host(c_slot).reshape(B, n), host(c_sent).reshape(B, n + 1, L)
            par, tk = host(parent).reshape(B, n), host(tok).reshape(B, n)
            for b in range(B):
                heap = rnd["partial"][b]
                assert pc[b] == len(heap), (n, it, b)
                for j, bm in enumerate(heap):
                    assert sn[b, j, :pl[b, j]].tolist() == bm["sentence"] and ps[b, j] == bm["score"] and plp[b, j] == bm["logprob"], (n, it, b, j)
                    assert par[b, j] == b * n + bm["parent"] and tk[b, j] == bm["sentence"][-1]
                cheap = rnd["complete"][b]
                assert cc[b] == len(cheap), (n, it, b)
                for j, bm in enumerate(cheap):
                    assert cst[b, csl[b, j], :cl[b, j]].tolist() == bm["sentence"] and cs[b, j] == bm["score"] and clp[b, j] == bm["logprob"]


@pytest.mark.parametrize("beam,max_len,check_every", [(10, 9, 4), (1, 12, 4), (3, 3, 4), (3, 12, 3), (8, 7, 0), (2, 6, 2)],
                         ids=["beam10-unfused-topk", "beam1", "two-rounds-no-graph", "odd-check-interval", "beam8-no-checks", "chunks-of-two"])
def test_beam_search_edge_shapes_match_the_oracle_eager_and_replayed(lib, beam, max_len, check_every, monkeypatch):
    """the round-6 decode paths at their edges: beam > 8 (separate softmax + k-pass top-k), beam 1, fewer rounds than a chunk (never
    captured), a check interval that is not a chunk size, no host checks at all, chunks of two rounds -- each called TWICE (eager +
    capture, then replay) and with VC_DECODE_XPROJ=0, always the fp64 oracle's beams (vae_model/decoder.py:203-320)."""
    p, eng, gen, P64, feats, cv, eps, cm = setup(lib, 11, prior="GMM")
    ref = [od.beam_search(P64, p, feats[b].astype(np.float64), cv[b].astype(np.float64), eps[:, b:b + 1].astype(np.float64), BOS, EOS,
                          c_means=cm, beam_size=beam, max_len=max_len)[0] for b in range(feats.shape[0])]
    for xproj in ("1", "0"):
        monkeypatch.setenv("VC_DECODE_XPROJ", xproj)
        g = CaptionGenerator(eng)
        for call in range(2):
            got = g.beam_search(feats, None, eps, BOS, EOS, beam_size=beam, max_len=max_len, check_every=check_every)
            assert [[s for s, _ in got[b]] for b in range(len(got))] == ref, (xproj, call)


@pytest.mark.parametrize("slices", [2, 3])
def test_sliced_beam_search_on_streams_returns_the_single_slice_beams(lib, slices):
    """A batch of >= 2 x 256 rows is decoded as two slices of images on two streams (generate.py beam_search).  Here the row
    threshold is lowered so that six images run as 2 x 3 and 3 x 2: same beams and scores as one slice and as the fp64 oracle, on the
    eager first call, on the replayed second, and after a call of another width in between (buffers and graphs are per slice shape)."""
    p, eng, gen, P64, feats, cv, eps, cm = setup(lib, 13, prior="GMM")
    ref = [od.beam_search(P64, p, feats[b].astype(np.float64), cv[b].astype(np.float64), eps[:, b:b + 1].astype(np.float64), BOS, EOS,
                          c_means=cm, beam_size=4, max_len=12)[0] for b in range(feats.shape[0])]
    one = CaptionGenerator(eng)
    one.slices = 1
    single = one.beam_search(feats, None, eps, BOS, EOS, beam_size=4, max_len=12)
    g = CaptionGenerator(eng)
    g.slices, g.slice_rows = slices, 1
    for call in range(3):
        got = g.beam_search(feats, None, eps, BOS, EOS, beam_size=4, max_len=12)
        assert len(g._side) == slices - 1
        assert [[s for s, _ in r] for r in got] == ref, call
        assert [[s for s, _ in r] for r in got] == [[s for s, _ in r] for r in single]
        np.testing.assert_allclose([sc for r in got for _, sc in r], [sc for r in single for _, sc in r], rtol=1e-5, atol=1e-6)
        if call == 0:
            g.beam_search(feats, None, eps, BOS, EOS, beam_size=2, max_len=12)


@pytest.mark.parametrize("max_len,check_every", [(1, 4), (3, 4), (9, 0), (10, 2), (13, 3)])
def test_greedy_edge_lengths_match_the_oracle(lib, max_len, check_every):
    p, eng, gen, P64, feats, cv, eps, cm = setup(lib, 13, no_encoder=True)
    for call in range(2):
        got = gen.greedy(feats, None, None, BOS, EOS, max_len=max_len, check_every=check_every)
        for b in range(feats.shape[0]):
            ref = od.greedy(P64, p, feats[b].astype(np.float64), cv[b].astype(np.float64), None, BOS, EOS, c_means=cm, max_len=max_len)
            assert got[b] == ref, (call, b, got[b], ref)
