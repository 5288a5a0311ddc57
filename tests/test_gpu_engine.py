"""-m gpu: the whole caption-side training step (forward, every gradient, clip + optimiser,
3 consecutive steps) through the C ABI vs the CPU oracle on identical injected tensors.

Tolerance (north_star): per-step loss / KL within 1e-3 absolute in fp32; asserted here
at 2e-4 relative, gradients at 2e-4 of each tensor's max, parameters after 3 steps at
1e-4 of the total update size."""
import numpy as np
import pytest
import torch

from oracle import caption_model as cm
from oracle import decode as odec
from oracle import optim as oo
from vae_captioning_amd import spec, synth
from vae_captioning_amd.engine import CaptionEngine
from vae_captioning_amd.utils.parameters import Parameters

from .gpu_util import grads_only

pytestmark = pytest.mark.gpu


def small_params(**kw):
    p = Parameters()
    p.embed_size, p.encoder_hidden, p.decoder_hidden = 32, 64, 96
    p.latent_size, p.gen_z_samples, p.cnn_feature_size = 12, 5, 40
    p.num_captions, p.batch_size = 3, 4
    p.lstm_clip_by_norm = 0.05  # small enough that the clip is active
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def make_case(p, V, B, T, seed):
    rng = np.random.default_rng(seed)
    P = spec.init_caption_params(p, V, seed=seed + 1)
    for k in P:
        if k.endswith("bias"):
            P[k] = rng.normal(0, 0.1, P[k].shape).astype(np.float32)
    batch = synth.make_batch(rng, B, p.num_captions, T, V, use_ci=spec.uses_ci(p), variable_len=True,
                             feature_size=p.cnn_feature_size)
    noise = synth.make_noise(rng, B * p.num_captions, T, p)
    return P, batch, noise


def oracle_steps(p, P, batch, noise, nsteps, optimizer="Adam", dtype=np.float64):
    P = {k: v.astype(dtype) for k, v in P.items()}
    b = {k: (v.astype(dtype) if v.dtype.kind == "f" else v) for k, v in batch.items()}
    n = {k: (v.astype(dtype) if v.dtype.kind == "f" else v) for k, v in noise.items()}
    if p.prior == "AG":
        n["c_means"] = odec.init_clusters(90, p.latent_size).astype(dtype)
    st = {}
    hist = []
    first = None
    for s in range(nsteps):
        out = cm.forward_backward(P, b, n, p, global_step=s)
        norm = oo.global_norm({k: v.astype(np.float32) for k, v in out.grads.items()},
                              {k: v.astype(np.float32) for k, v in out.sparse.items()})
        scale = float(np.float64(p.lstm_clip_by_norm) * min(1.0 / float(norm), 1.0 / p.lstm_clip_by_norm))
        hist.append((float(np.mean(out.kld)), float(out.rec_loss), float(np.mean(out.lower_bound)), float(norm)))
        if first is None:
            first = out
        g32 = {k: v.astype(np.float32) for k, v in out.grads.items()}
        P32 = {k: v.astype(np.float32) for k, v in P.items()}
        if optimizer == "Adam":
            oo.adam_step(P32, g32, st, p.learning_rate, s + 1, scale=scale)
        elif optimizer == "SGD":
            oo.sgd_step(P32, g32, oo.decayed_lr(p.learning_rate, s, p.num_ex_per_epoch, p.batch_size, p.num_epochs_per_decay), scale=scale)
        else:
            touched = {"decoder/net/dec_embeddings": np.isin(np.arange(P32["decoder/net/dec_embeddings"].shape[0]), batch["cap_dec"])}
            if "encoder/enc_embeddings" in P32:
                touched["encoder/enc_embeddings"] = np.isin(np.arange(P32["encoder/enc_embeddings"].shape[0]), batch["cap_enc"])
            oo.momentum_step(P32, g32, st, oo.decayed_lr(p.learning_rate, s, p.num_ex_per_epoch, p.batch_size, p.num_epochs_per_decay),
                             scale=scale, touched=touched)
        P = {k: v.astype(dtype) for k, v in P32.items()}
    return first, hist, P


VARIANTS = [
    dict(prior="Normal"),
    dict(prior="Normal", no_encoder=True),
    dict(prior="Normal", use_c_v=True, dec_keep_rate=0.8, dec_lstm_drop=0.7),
    dict(prior="GMM"),
    dict(prior="AG", use_c_v=True),
    dict(prior="AG"),
    dict(prior="Normal", ann_param=2.0, optimizer="SGD"),
    dict(prior="Normal", optimizer="Momentum"),
]


@pytest.mark.parametrize("kw", VARIANTS, ids=lambda k: "-".join("%s=%s" % i for i in k.items()))
def test_train_steps_match_oracle(lib, kw):
    p = small_params(**kw)
    V, B, T = 203, 4, 6
    P0, batch, noise = make_case(p, V, B, T, seed=7)
    first, hist, Pref = oracle_steps(p, P0, batch, noise, 3, optimizer=p.optimizer)

    eng = CaptionEngine(p, V, lib=lib)
    eng.load_params(P0)
    nsteps = 3
    for s in range(nsteps):
        eng.set_batch(batch, noise)
        eng.forward()
        eng.backward()
        eng.pack_tail()
        if s == 0:
            G = eng.grads_dict()
            for name, ref in first.grads.items():
                tol = 2e-4 * (np.abs(ref).max() + 1e-12)
                err = np.abs(G[name] - ref).max()
                assert err <= tol, "grad %s: err %.3e tol %.3e (max %.3e)" % (name, err, tol, np.abs(ref).max())
        eng.apply_gradients()
        kld, rec, lb, ann = eng.losses()
        rk, rr, rl, rn = hist[s]
        assert abs(rec - rr) <= 2e-4 * abs(rr), ("rec_loss step %d" % s, rec, rr)
        assert abs(kld - rk) <= 2e-4 * abs(rk) + 1e-6, ("kld step %d" % s, kld, rk)
        assert abs(lb - rl) <= 2e-4 * abs(rl), ("lower_bound step %d" % s, lb, rl)
        norm = float(eng.ns[0].item())
        assert abs(norm - rn) <= 3e-4 * rn, ("global norm step %d" % s, norm, rn)
        assert rn > p.lstm_clip_by_norm, "clip not active in this case"
    Pg = eng.state_dict()
    for name, ref in Pref.items():
        upd = np.abs(ref - P0[name]).max()
        err = np.abs(Pg[name] - ref).max()
        assert err <= 2e-3 * upd + 1e-7, "param %s after %d steps: err %.3e, update size %.3e" % (name, nsteps, err, upd)
    assert int(eng.step.item()) == nsteps


def test_eval_forward_does_not_touch_state(lib):
    p = small_params(prior="Normal")
    V, B, T = 203, 4, 6
    P0, batch, noise = make_case(p, V, B, T, seed=9)
    eng = CaptionEngine(p, V, lib=lib)
    eng.load_params(P0)
    eng.set_batch(batch, noise)
    eng.forward(train=False)
    _, rec, _, _ = eng.losses()
    out = cm.forward_backward({k: v.astype(np.float64) for k, v in P0.items()},
                              {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in batch.items()},
                              {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in noise.items()}, p, want_grads=False)
    assert abs(rec - float(out.rec_loss)) <= 2e-4 * abs(float(out.rec_loss))
    assert int(eng.step.item()) == 0
    Pg = eng.state_dict()
    for k in P0:
        np.testing.assert_array_equal(Pg[k], P0[k])


def test_device_noise_runs_and_is_fresh_each_step(lib):
    p = small_params(prior="Normal", dec_keep_rate=0.9)
    V, B, T = 203, 4, 6
    P0, batch, _ = make_case(p, V, B, T, seed=11)
    eng = CaptionEngine(p, V, lib=lib, seed=3)
    eng.load_params(P0)
    eng.set_batch(batch)
    eng.forward(); eng.backward(); eng.pack_tail(); eng.apply_gradients()
    e1 = eng.buf["eps"].clone()
    l1 = eng.losses()
    eng.forward(); eng.backward(); eng.pack_tail(); eng.apply_gradients()
    assert not torch.equal(e1, eng.buf["eps"])
    assert all(np.isfinite(l1)) and all(np.isfinite(eng.losses()))
    assert abs(float(e1.mean())) < 0.1 and abs(float(e1.std()) - 1) < 0.1


def test_bucketed_async_allreduce_equals_single_allreduce_on_rccl(lib):
    """Fine-tune step on a 1-rank nccl group: the three overlapped all-reduce pieces (caption | fc | conv)
    give bit-identical parameters to the single blocking all-reduce and to the collective-free step."""
    import os
    import torch.distributed as dist
    from vae_captioning_amd.trainer import Trainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        p = Parameters()
        p.fine_tune, p.batch_size, p.num_captions, p.gen_z_samples = True, 2, 2, 4
        V = 300
        rng = np.random.default_rng(4)
        batch = synth.make_batch(rng, 2, 2, 5, V, images=True, variable_len=True)
        P0 = {**spec.init_caption_params(p, V, seed=1), **spec.init_vgg_params(seed=3)}
        res = []
        for force, buckets in ((False, False), (True, False), (True, True)):
            tr = Trainer(p, V, lib=lib, force_collectives=force, seed=5, comm="torch")   # (the libvaecap communicator: tests/test_gpu_comm.py)
            tr.buckets = buckets
            tr.load_state_dict(P0)
            tr.set_batch(batch)
            tr.train_step()
            res.append((tr.losses(), grads_only(tr)))
        assert res[0][0] == res[1][0] == res[2][0]
        assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][1], res[2][1])
    finally:
        dist.destroy_process_group()


def test_collective_code_path_on_a_one_rank_rccl_group(lib):
    """The data-parallel branches (count all-reduce, loss-scalar all-reduce, the single flat
    gradient all-reduce over RCCL) executed on a 1-rank nccl group: results must equal the
    collective-free step bit for bit (sum over one rank is the identity)."""
    import os
    import torch.distributed as dist
    from vae_captioning_amd.trainer import Trainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        p = small_params(prior="Normal")
        V, B, T = 203, 4, 6
        P0, batch, noise = make_case(p, V, B, T, seed=13)
        res = []
        for force in (False, True):
            tr = Trainer(p, V, lib=lib, force_collectives=force, comm="torch")
            tr.load_state_dict(P0)
            for _ in range(2):
                tr.set_batch(batch, noise)
                tr.train_step()
            res.append((tr.losses(), tr.state_dict()))
        assert res[0][0] == res[1][0]
        for k in res[0][1]:  # every kernel is deterministic (no atomics on the training path)
            np.testing.assert_array_equal(res[0][1][k], res[1][1][k])
    finally:
        dist.destroy_process_group()


def test_gmm_component_is_drawn_on_device_from_softmax_of_the_cluster_vector(lib):
    """encoder.py:72-75 (quirk Q15): tf.multinomial(c_i_ph, 1) uses the cluster vector as LOGITS.  Without injected noise the
    engine draws the component on the device (Philox uniform + inverse CDF): in range, reproducible for a given step, fresh on
    the next step, and distributed as softmax(c_v)."""
    p = small_params(prior="GMM")
    p.num_captions, p.batch_size = 1, 4000
    V, B, T = 60, 4000, 4
    rng = np.random.default_rng(0)
    batch = synth.make_batch(rng, B, 1, T, V, use_ci=True, feature_size=p.cnn_feature_size)
    cv = np.zeros((B, 90), np.float32)
    cv[:, 3], cv[:, 17] = 2.0, 1.0                       # every row the same logits: frequencies estimate softmax(c_v)
    batch["c_v"] = cv
    e = CaptionEngine(p, V, lib=lib, seed=5)
    e.load_params(spec.init_caption_params(p, V, seed=1))
    e.set_batch(batch)                                   # no noise injected
    draws = []
    for s in (0, 0, 1):
        e.step.fill_(s)
        e.fw_prepare(train=False)
        draws.append(e.buf["gmm_idx"].cpu().numpy().copy())
    assert draws[0].min() >= 0 and draws[0].max() < 90
    np.testing.assert_array_equal(draws[0], draws[1])
    assert (draws[0] != draws[2]).mean() > 0.5
    pr = np.exp(cv[0].astype(np.float64)); pr /= pr.sum()
    freq = np.bincount(draws[0], minlength=90) / B
    assert np.abs(freq - pr).max() < 0.03, (freq[[3, 17]], pr[[3, 17]])
    e.forward(train=True)                                # and the step runs end to end with the drawn components
    assert np.isfinite(e.out.cpu().numpy()).all()


def _run_sequence(lib, p, V, P0, batches, synchronous, seed=7):
    """`len(batches)` training steps on a fresh engine; synchronous=True drains the device after every call (no upload can
    overlap a step), False leaves the two-stream pipeline of set_batch alone."""
    eng = CaptionEngine(p, V, lib=lib, seed=seed)
    eng.load_params(P0)
    losses = []
    for b in batches:
        eng.set_batch(b)
        if synchronous:
            torch.cuda.synchronize()
        eng.forward(); eng.backward(); eng.pack_tail(); eng.apply_gradients()
        if synchronous:
            torch.cuda.synchronize()
            losses.append(eng.losses())
    if not synchronous:
        losses.append(eng.losses())
    return losses, eng.state_dict(), eng


def test_caption_length_changes_between_steps_equal_the_synchronous_upload(lib):
    """Batch_Generator pads every batch to ITS longest caption, so T changes almost every step.  The staging / landing / index
    buffers are sized to a high-water mark (engine._upload_pack): T_k == T_{k+2} != T_{k+1}, growth in the middle of the run and a
    return to the small size must give bit-identical parameters to a run that drains the device around every call."""
    p = small_params(prior="Normal", dec_keep_rate=0.9)
    V, B = 203, 6
    rng = np.random.default_rng(21)
    P0 = spec.init_caption_params(p, V, seed=2)
    Ts = [9, 6, 9, 6, 9, 14, 5, 14, 9, 6, 23, 6]
    batches = [synth.make_batch(rng, B, p.num_captions, T, V, variable_len=True, feature_size=p.cnn_feature_size) for T in Ts]
    la, Pa, ea = _run_sequence(lib, p, V, P0, batches, synchronous=False)
    ls, Ps, _ = _run_sequence(lib, p, V, P0, batches, synchronous=True)
    assert la[-1] == ls[-1]
    for k in Pa:
        np.testing.assert_array_equal(Pa[k], Ps[k], err_msg=k)
    st = ea.pinned["__pack__"]
    assert st["cap"] >= max(ea.buf[k].numel() * 4 for k in ("cap_dec_t", "cap_enc_t"))
    assert len({d.data_ptr() for d in st["devs"]}) == 2 and not ea.retired   # losses() released what growth replaced


def test_vocabulary_beyond_the_device_scan_builds_the_index_on_the_host(lib):
    """vc_embedding_grad_index scans one LDS table (<= vc_embedding_index_max_vocab ids); larger vocabularies ship the host-built
    index in the same upload: one training step equals the oracle."""
    p = small_params(prior="Normal")
    p.embed_size = 8
    V = int(lib.vc_embedding_index_max_vocab()) + 37
    B, T = 4, 5
    P0, batch, noise = make_case(p, V, B, T, seed=5)
    eng = CaptionEngine(p, V, lib=lib)
    eng.load_params(P0)
    eng.set_batch(batch, noise)
    eng.forward(); eng.backward(); eng.pack_tail()
    first, hist, _ = oracle_steps(p, P0, batch, noise, 1)
    kld, rec, lb, ann = eng.losses()
    assert abs(rec - hist[0][1]) <= 2e-4 * abs(hist[0][1])
    G = eng.grads_dict()
    for name in ("decoder/net/dec_embeddings", "encoder/enc_embeddings"):
        ref = first.grads[name]
        assert np.abs(G[name] - ref).max() <= 2e-4 * np.abs(ref).max(), name


def test_set_batch_under_a_captured_graph_feeds_the_replayed_step(lib):
    """Trainer.capture() makes the inputs persistent tensors (engine.fix_inputs): a later set_batch copies INTO them, so a replay
    trains on the new batch -- same parameters as the eager run over the same batches -- and a batch of another shape is refused."""
    from vae_captioning_amd.trainer import Trainer
    p = small_params(prior="Normal")
    V, B, T = 203, 4, 6
    rng = np.random.default_rng(31)
    P0 = spec.init_caption_params(p, V, seed=4)
    batches = [synth.make_batch(rng, B, p.num_captions, T, V, variable_len=True, feature_size=p.cnn_feature_size) for _ in range(4)]
    res = []
    for graph in (False, True):
        tr = Trainer(p, V, lib=lib, seed=9)
        tr.load_state_dict(P0)
        tr.set_batch(batches[0])
        if graph:
            tr.capture(warmup=1)
            tr.cap.step.zero_()      # the warm-up ran the step once on batch 0: rewind to the same starting point
            tr.load_state_dict(P0)
            for s in tr.cap.store.slots.values():
                s.zero_()
        for b in batches:
            tr.set_batch(b)
            tr.train_step()
        res.append((tr.losses(), tr.state_dict()))
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        np.testing.assert_array_equal(res[0][1][k], res[1][1][k], err_msg=k)
    with pytest.raises(ValueError):
        tr.set_batch(synth.make_batch(rng, B, p.num_captions, T + 2, V, feature_size=p.cnn_feature_size))


@pytest.mark.parametrize("kw", [dict(), dict(prior="GMM"), dict(prior="AG", use_c_v=True), dict(no_encoder=True), dict(fine_tune=True),
                                dict(fine_tune=True, collectives=True), dict(collectives=True), dict(graph=True), dict(fine_tune=True, graph=True)],
                         ids=["normal", "gmm", "ag_cv", "no_encoder", "fine_tune", "fine_tune_rccl_buckets", "rccl", "hipgraph", "fine_tune_hipgraph"])
def test_weight_gradient_stream_equals_program_order_bit_for_bit(lib, kw):
    """Trainer's second stream (weight gradients of the caption side, clip + optimiser, fc1 / fc2's optimiser; engine.off_chain)
    against the same step in program order on one stream: the same kernels on the same data, so losses, gradients and updated
    parameters of three steps are identical -- a missing stream dependency would show as a difference (or as garbage).
    (fine_tune_hipgraph: a captured fine-tune step; Trainer.capture runs the VGG16 on one stream, see VggEngine.one_stream.)"""
    from vae_captioning_amd.trainer import Trainer
    kw = dict(kw)
    coll, graph = kw.pop("collectives", False), kw.pop("graph", False)
    p = Parameters()
    for k, v in kw.items():
        setattr(p, k, v)
    V, T, B = 600, 11, 6
    p.batch_size = B
    rng = np.random.default_rng(8)
    batch = synth.make_batch(rng, B, p.num_captions, T, V, use_ci=spec.uses_ci(p), variable_len=True, images=bool(p.fine_tune))
    P0 = spec.init_caption_params(p, V, seed=1)
    if p.fine_tune:
        P0.update(spec.init_vgg_params(seed=2))
    res = []
    for side in (True, False):
        tr = Trainer(p, V, lib=lib, seed=5, force_collectives=coll, wgrad_stream=side)
        assert (tr.cap.wgrad_stream is not None) == side
        tr.load_state_dict(P0)
        tr.set_batch(batch)
        if graph:
            tr.capture()
        for _ in range(3):
            tr.train_step()
        torch.cuda.synchronize()
        res.append((tr.losses(), tr.gall.clone(), tr.cap.store.p.clone(), tr.vgg.store.p.clone() if tr.vgg is not None else None))
        if tr.comm is not None:
            tr.comm.destroy()
        del tr
    assert res[0][0] == res[1][0] and all(np.isfinite(res[0][0])), (res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1]) and float(res[0][1].abs().max()) > 0
    assert torch.equal(res[0][2], res[1][2])
    if res[0][3] is not None:
        assert torch.equal(res[0][3], res[1][3])
