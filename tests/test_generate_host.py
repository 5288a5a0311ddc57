"""CPU: the host end of beam search -- `generate.beams_from_host` turns a slice's result buffers into per-image lists of
(sentence, score) -- against the straightforward reading of vae_model/decoder.py:295-320 with this build's Beam / TopN classes
(themselves pinned to the reference's utils/top_n.py by tests/test_ref_fixtures.py): complete captions if an image has any, else
its live beams, never mixed; `TopN.extract(sort=True)` order (list.sort(reverse=True) on the heap ARRAY: descending score, equal
scores in array order).  Random buffers with many exact score ties, images with / without complete captions, pool slots in any
order, both sentence parities."""
import numpy as np
import pytest

from vae_captioning_amd.generate import beams_from_host
from vae_captioning_amd.utils.top_n import Beam, TopN


def _buffers(rng, B, n, L, complete):
    M = B * n
    sizes = [("pcount", B), ("ccount", B), ("p_len", M), ("c_len", M), ("c_slot", M), ("sent0", M * L), ("sent1", M * L), ("c_sent", B * (n + 1) * L)]
    io, o = {}, 0
    for name, sz in sizes:
        io[name] = o
        o += sz
    ints = rng.integers(3, 5000, size=o).astype(np.int32)          # (every token position holds junk: only [:len] may be read)
    f = lambda name, cnt: ints[io[name]:io[name] + cnt]
    f("pcount", B)[:] = rng.integers(0, n + 1, size=B)
    cc = rng.integers(1, n + 1, size=B) if complete == "all" else (np.zeros(B, np.int64) if complete == "none" else rng.integers(0, n + 1, size=B))
    f("ccount", B)[:] = cc
    f("p_len", M)[:] = rng.integers(1, L + 1, size=M)
    f("c_len", M)[:] = rng.integers(1, L + 1, size=M)
    f("c_slot", M)[:] = np.concatenate([rng.permutation(n + 1)[:n] for _ in range(B)])   # distinct pool rows per image, any order
    dbls = rng.choice(np.array([-0.5, -1.25, -1.25, -2.0, -3.5]), size=2 * M)             # few distinct scores: ties in every image
    return ints, dbls, io


def _reference(ints, dbls, io, B, n, L, last):
    M = B * n
    f = lambda name, cnt, shape: ints[io[name]:io[name] + cnt].reshape(shape)
    pc, cc = f("pcount", B, (B,)), f("ccount", B, (B,))
    pl, cl, csl = f("p_len", M, (B, n)), f("c_len", M, (B, n)), f("c_slot", M, (B, n))
    psent, csent = f("sent%d" % last, M * L, (B, n, L)), f("c_sent", B * (n + 1) * L, (B, n + 1, L))
    ps, cs = dbls[:M].reshape(B, n), dbls[M:].reshape(B, n)
    out = []
    for b in range(B):
        complete, partial = TopN(n), TopN(n)
        complete._heap = [Beam(csent[b, csl[b, j], :cl[b, j]].tolist(), None, None, float(cs[b, j])) for j in range(cc[b])]   # heap ARRAY order
        partial._heap = [Beam(psent[b, j, :pl[b, j]].tolist(), None, None, float(ps[b, j])) for j in range(pc[b])]
        if not complete.size():          # decoder.py:295-299
            complete = partial
        out.append([(bm.sentence, bm.score) for bm in complete.extract(sort=True)])
    return out


@pytest.mark.parametrize("complete", ["none", "all", "mixed"])
@pytest.mark.parametrize("B,n,L,last", [(1, 1, 4, 0), (5, 3, 9, 1), (64, 5, 32, 0), (7, 8, 12, 1), (3, 16, 6, 0)])
def test_result_buffers_become_the_lists_the_reference_returns(B, n, L, last, complete):
    rng = np.random.default_rng(B * 100 + n + last)
    ints, dbls, io = _buffers(rng, B, n, L, complete)
    got = beams_from_host(ints, dbls, io, B, n, L, last)
    ref = _reference(ints, dbls, io, B, n, L, last)
    assert got == ref
    assert all(isinstance(t, int) for img in got for s, _ in img for t in s) and all(isinstance(sc, float) for img in got for _, sc in img)
