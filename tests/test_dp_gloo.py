"""CPU, world_size 2, gloo: the data-parallel protocol of vae_captioning_amd/dp.py
(shard by image, count all-reduce, ONE flat-gradient all-reduce with the reduction scalars in
its tail, identical optimiser step on each replica).  The per-rank compute is the oracle
(there is no GPU here); the expected result is the single-process oracle on the global batch
with q1_groups = 2 (each rank mixes z samples inside its own shard, see dp.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(prior, use_c_v):
    sys.path.insert(0, ROOT)
    from oracle import caption_model as cm
    from oracle import decode
    from vae_captioning_amd import spec, synth
    cfg = cm.default_cfg(embed_size=8, encoder_hidden=12, decoder_hidden=12, latent_size=6, gen_z_samples=3,
                         num_captions=2, cnn_feature_size=20, vocab_size=31, prior=prior, use_c_v=use_c_v)
    rng = np.random.default_rng(3)
    P = {k: v.astype(np.float64) for k, v in spec.init_caption_params(cfg, 31, seed=4).items()}
    batch = synth.make_batch(rng, 4, 2, 5, 31, use_ci=spec.uses_ci(cfg), variable_len=True, feature_size=20)
    batch["features"] = batch["features"].astype(np.float64)
    if "c_v" in batch:
        batch["c_v"] = batch["c_v"].astype(np.float64)
    noise = synth.make_noise(rng, 8, 5, cfg)
    noise = {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in noise.items()}
    if prior == "AG":
        noise["c_means"] = decode.init_clusters(90, 6).astype(np.float64)
    return cfg, P, batch, noise


def _worker(rank, world, port, prior, use_c_v, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from oracle import caption_model as cm
        from oracle import optim as oo
        from vae_captioning_amd import dp
        cfg, P, batch, noise = _case(prior, use_c_v)
        nc = cfg.num_captions
        n_glob = batch["cap_dec"].shape[0]
        b = dp.shard_batch(batch, rank, world, nc)
        n = dp.shard_noise(noise, rank, world, n_glob)
        # (1) count all-reduce
        den = torch.tensor([float((b["cap_enc"] != 0).sum())], dtype=torch.float64)
        dist.all_reduce(den)
        # (2) local forward/backward with the shard scales
        out = cm.forward_backward(P, b, n, cfg, global_step=0, dp=dict(ce_den=float(den), n_rows=n_glob))
        names = sorted(out.grads)
        dx_sq = sum(float((v ** 2).sum()) for v in out.sparse.values())
        kl_sum = float(np.sum(out.kld)) if prior == "AG" else float(out.kld)
        flat = np.concatenate([out.grads[k].ravel() for k in names] + [np.array([dx_sq, float(out.ce_num), kl_sum])])
        t = torch.from_numpy(flat)
        # (3) the single gradient all-reduce
        dist.all_reduce(t)
        flat = t.numpy()
        G, off = {}, 0
        for k in names:
            sz = out.grads[k].size
            G[k] = flat[off:off + sz].reshape(out.grads[k].shape)
            off += sz
        dx_sq, ce_num, kl_sum = flat[off:off + 3]
        # (4) clip + Adam on every replica
        dense_sq = sum(float((G[k] ** 2).sum()) for k in names if k not in out.sparse)
        norm = np.sqrt(dense_sq + dx_sq)
        scale = 5.0 * min(1.0 / norm, 1.0 / 5.0)
        P32 = {k: v.astype(np.float32) for k, v in P.items()}
        oo.adam_step(P32, {k: G[k].astype(np.float32) for k in names}, {}, 5e-4, 1, scale=scale)
        if rank == 0:
            ret["G"] = {k: G[k].copy() for k in names}
            ret["rec"] = ce_num / float(den)
            ret["kl"] = kl_sum / (n_glob if prior == "AG" else 1.0)
            ret["norm"] = norm
            ret["P"] = P32
        chk = torch.tensor([float(sum(v.astype(np.float64).sum() for v in P32.values()))], dtype=torch.float64)
        both = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(both, chk)
        assert all(float(x) == float(both[0]) for x in both), "replicas diverged"
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("prior,use_c_v", [("Normal", False), ("AG", True), ("GMM", False)])
def test_two_rank_step_equals_global_batch_oracle(prior, use_c_v):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), prior, use_c_v, ret), nprocs=2, join=True)
    sys.path.insert(0, ROOT)
    from oracle import caption_model as cm
    from oracle import optim as oo
    cfg, P, batch, noise = _case(prior, use_c_v)
    ref = cm.forward_backward(P, batch, noise, cfg, global_step=0, q1_groups=2)
    np.testing.assert_allclose(ret["rec"], float(ref.rec_loss), rtol=1e-12)
    np.testing.assert_allclose(ret["kl"], float(np.mean(ref.kld)), rtol=1e-10)
    for k, g in ref.grads.items():
        np.testing.assert_allclose(ret["G"][k], g, rtol=1e-9, atol=1e-13, err_msg=k)
    norm = float(oo.global_norm({k: v for k, v in ref.grads.items()}, ref.sparse))
    np.testing.assert_allclose(ret["norm"], norm, rtol=1e-5)
    # and it is NOT the single-GPU Q1 mix of the concatenated batch (documented difference)
    one = cm.forward_backward(P, batch, noise, cfg, global_step=0, q1_groups=1)
    assert abs(float(one.rec_loss) - float(ref.rec_loss)) > 0


def test_shard_batch_keeps_an_images_rows_together():
    from vae_captioning_amd import dp
    B, nc = 6, 5
    batch = dict(features=np.arange(B)[:, None].astype(np.float32), cap_dec=np.repeat(np.arange(B), nc)[:, None],
                 cap_enc=np.repeat(np.arange(B), nc)[:, None], lengths=np.repeat(np.arange(B), nc))
    for r in range(3):
        s = dp.shard_batch(batch, r, 3, nc)
        assert s["features"][:, 0].tolist() == [2 * r, 2 * r + 1]
        assert s["lengths"].tolist() == [2 * r] * nc + [2 * r + 1] * nc
    g, kn, ka, inv = dp.scales(320, 8, True)
    assert g == 2560.0 and abs(kn - 0.1 / 2560) < 1e-12 and ka == 0.1 and inv == 1 / 2560
    assert dp.scales(320, 8, False)[0] == 1.0


def test_global_q1_noise_slices_tile_the_global_sample_tensor():
    """q1_mode='global': rank r owns the flat range [r*Nl*S, (r+1)*Nl*S) of q = s*Ng + n, i.e. exactly the
    rows r*Nl .. (r+1)*Nl of the reference's [Ng, S*L] reshape (vae_model/decoder.py:109-110)."""
    from oracle import ops
    from vae_captioning_amd import dp
    S, Ng, L, world = 5, 12, 3, 4
    eps = np.random.default_rng(0).standard_normal((S, Ng, L)).astype(np.float32)
    zin = ops.q1_reshape(eps, L, S)  # [Ng, S*L]
    nl = Ng // world
    for r in range(world):
        e = dp.shard_noise({"eps": eps}, r, world, Ng, "global")["eps"]
        assert e.shape == (S, nl, L)
        np.testing.assert_array_equal(e.reshape(nl, S * L), zin[r * nl:(r + 1) * nl])
        t = dp.shard_noise({"eps": eps}, r, world, Ng, "tower")["eps"]
        np.testing.assert_array_equal(t, eps[:, r * nl:(r + 1) * nl])


def test_gradient_buckets_are_disjoint_and_cover_the_flat_buffer():
    """The ONE logical all-reduce of `gall` is issued as four pieces with VGG fine-tuning (trainer.Trainer._step): the slices
    must be disjoint, cover [0, len(gall)) exactly, and each VGG piece must hold exactly the layers whose gradients are final
    when it is issued (fc1/fc2 first, then conv3_1..conv5_3, then conv1_1..conv2_2)."""
    from vae_captioning_amd import dp, spec
    from vae_captioning_amd.engine import TAIL, flat_offsets, internal_caption_variables
    from vae_captioning_amd.utils.parameters import Parameters
    p = Parameters()
    p.fine_tune = True
    _, n_cap = flat_offsets(internal_caption_variables(p, 10000))
    n_cap += TAIL
    voff, n_vgg = flat_offsets(spec.vgg_variables())
    off_fc, off_c3 = n_cap + voff["cnn/fc1/weights"][0], n_cap + voff["cnn/conv3_1/weights"][0]
    bk = dp.gradient_buckets(n_cap, n_cap + n_vgg, off_fc, off_c3)
    assert len(bk) == 4 and all(lo < hi for lo, hi in bk)
    srt = sorted(bk)
    assert srt[0][0] == 0 and srt[-1][1] == n_cap + n_vgg
    assert all(a[1] == b[0] for a, b in zip(srt, srt[1:]))          # no gap, no overlap
    inside = lambda name, b: b[0] <= n_cap + voff[name][0] and n_cap + voff[name][0] + int(np.prod(voff[name][1])) <= b[1]
    assert bk[0] == (0, n_cap)
    assert inside("cnn/fc1/weights", bk[1]) and inside("cnn/fc2/biases", bk[1])
    assert inside("cnn/conv3_1/weights", bk[2]) and inside("cnn/conv5_3/biases_conv", bk[2]) and not inside("cnn/conv2_2/weights", bk[2])
    assert inside("cnn/conv1_1/weights", bk[3]) and inside("cnn/conv2_2/biases", bk[3])
    assert dp.gradient_buckets(n_cap, n_cap) == [(0, n_cap)]        # caption-only runs: the single all-reduce


def _handshake_worker(rank, world, port, case, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        import types
        from vae_captioning_amd import abi, dp
        from vae_captioning_amd.trainer import Trainer

        class Lib(object):   # the three entries the handshake touches, failing where the case says
            def vc_comm_available(self):
                return 0 if (case == "unavailable_on_1" and rank == 1) else 1

            def vc_comm_unique_id(self, buf):
                if case == "id_fails":
                    raise abi.VaecapError("vc_comm_unique_id: no RCCL (test)")
                buf.raw = bytes(range(128))

        made = []

        class FakeComm(object):
            def __init__(self, lib, world_, rank_, dev, uid):
                if case == "init_fails_on_0" and rank_ == 0:
                    raise abi.VaecapError("ncclCommInitRank failed (test)")
                assert bytes(uid) == bytes(range(128)) and world_ == world and rank_ == rank
                made.append(self)
                self.destroyed = False

            unique_id = staticmethod(dp.AbiComm.unique_id)

            def destroy(self):
                self.destroyed = True
        real = dp.AbiComm
        dp.AbiComm = FakeComm
        try:
            me = types.SimpleNamespace(lib=Lib(), world=world, rank=rank, group=None)
            comm = Trainer._agree_on_abi_comm(me, 0)
        finally:
            dp.AbiComm = real
        ret[rank] = (comm is not None, [c.destroyed for c in made])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["ok", "unavailable_on_1", "id_fails", "init_fails_on_0"])
def test_ranks_agree_on_the_communicator_whatever_fails_where(case):
    """Trainer._agree_on_abi_comm on two gloo ranks with the library's three entries stubbed: either both ranks get a communicator
    or both fall back to torch.distributed -- an RCCL that cannot be bound on one rank, a unique id that rank 0 cannot create, an
    init that fails on one rank only (the communicator the other rank did get is destroyed).  Nobody hangs."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_handshake_worker, args=(2, _free_port(), case, ret), nprocs=2, join=True)
    got = [ret[0], ret[1]]
    if case == "ok":
        assert got == [(True, [False]), (True, [False])]
    elif case == "init_fails_on_0":
        assert got == [(False, []), (False, [True])]
    else:
        assert got == [(False, []), (False, [])]
