"""-m gpu: the reference-named facades (Encoder / Decoder / optimizers / vgg16 / make_rnn_cell) and
the main.py command line (utils/parameters.py:75-132) drive the same engine as Trainer.train_step."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ops as O
from vae_captioning_amd import session, spec, synth
from vae_captioning_amd.ops import optimizers
from vae_captioning_amd.trainer import Trainer
from vae_captioning_amd.utils.parameters import Parameters
from vae_captioning_amd.utils.rnn_model import make_rnn_cell, rnn_placeholders
from vae_captioning_amd.vae_model.decoder import Decoder
from vae_captioning_amd.vae_model.encoder import Encoder

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params(**kw):
    p = Parameters()
    p.embed_size, p.encoder_hidden, p.decoder_hidden = 32, 64, 64
    p.latent_size, p.gen_z_samples, p.cnn_feature_size = 10, 4, 48
    p.num_captions, p.batch_size, p.vocab_size = 2, 3, 90
    for k, v in kw.items():
        setattr(p, k, v)
    return p


@pytest.mark.parametrize("kw", [dict(prior="Normal"), dict(prior="AG", use_c_v=True), dict(no_encoder=True)],
                         ids=["normal", "ag_cv", "lstm"])
def test_facade_step_equals_trainer_step(lib, kw):
    rng = np.random.default_rng(1)
    p1, p2 = _params(**kw), _params(**kw)
    P0 = spec.init_caption_params(p1, 90, seed=2)
    batch = synth.make_batch(rng, 3, 2, 6, 90, use_ci=spec.uses_ci(p1), variable_len=True, feature_size=48)
    noise = synth.make_noise(rng, 6, 6, p1)
    ref = Trainer(p1, 90, lib=lib)
    ref.load_state_dict(P0)
    ref.set_batch(batch, noise)
    ref.train_step()
    tr = session.get(p2)
    tr.load_state_dict(P0)
    tr.set_batch(batch, noise)
    cap = tr.cap
    enc = None if p2.no_encoder else Encoder(None, None, None, p2)
    dec = Decoder(None, None, None, p2, None)
    optimize, global_step, global_norm = optimizers.non_cnn_optimizer(None, p2)
    images_fv = cap.fw_prepare()
    assert tuple(images_fv.shape) == (6, 32)
    obs = {}
    if enc is not None:
        z, tm, tl = enc.q_net()
        assert tuple(z.shape) == (4, 6, 10)
        if p2.prior == "AG":
            assert tuple(tm.shape) == (6, 90, 10) and tuple(tl.shape) == (6, 90, 10)
        obs = {"z": z}
    _, x_logits, shpe, (init_state, final_state, sample) = dec.px_z_fi(obs)
    assert tuple(x_logits.shape) == (36, 90)
    cap.fw_loss()
    optimize()
    assert int(global_step.item()) == 1 and float(global_norm.item()) > 0
    assert tr.losses() == ref.losses()
    a, b = ref.state_dict(), tr.state_dict()
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])


@pytest.mark.parametrize("kw", [dict(prior="Normal"), dict(prior="AG", use_c_v=True), dict(no_encoder=True), dict(prior="Normal", fine_tune=True)],
                         ids=["normal", "ag_cv", "lstm", "fine_tune"])
def test_reference_call_sequence_with_arrays_as_arguments(lib, kw):
    """One training step driven ONLY through the reference's call sequence (main.py:74-78 vgg16, :84-94 imf_emb, :99-102 Encoder /
    Decoder, :108-116 cv_emb + c_i attributes, :117 q_net, :148-150 px_z_fi, :179-183 the optimiser functions) with the step's arrays
    passed as the constructor arguments the reference's placeholders occupy -- no Trainer.set_batch anywhere in this test.  Losses and
    every updated parameter equal Trainer.train_step on the same batch bit for bit; a second step with OTHER arrays trains on those,
    not on the resident batch."""
    from vae_captioning_amd import layers
    from vae_captioning_amd.utils.image_embeddings import vgg16
    rng = np.random.default_rng(3)
    fine = bool(kw.get("fine_tune"))
    p1, p2 = _params(**kw), _params(**kw)
    if fine:
        p1.cnn_feature_size = p2.cnn_feature_size = 4096
        p1.batch_size = p2.batch_size = 2
    B, nc, T, V = (2 if fine else 3), 2, 6, 90
    P0 = {**spec.init_caption_params(p1, V, seed=2), **(spec.init_vgg_params(seed=3) if fine else {})}
    batches = [synth.make_batch(rng, B, nc, T + i, V, use_ci=spec.uses_ci(p1), variable_len=True, feature_size=p1.cnn_feature_size, images=fine)
               for i in range(2)]
    ref = Trainer(p1, V, lib=lib, seed=11)
    ref.load_state_dict(P0)
    ref_losses = []
    for b in batches:
        ref.set_batch(b)
        ref.train_step()
        ref_losses.append(ref.losses())

    class CapDict(object):   # data.dictionary (main.py:91-92)
        vocab_size = V
    p2.vocab_size = CapDict.vocab_size
    tr = session.get(p2)             # the session object the facades share (the reference's graph + tf.Session)
    tr.cap.seed = 11
    if tr.vgg is not None:
        tr.vgg.seed = 11
    tr.load_state_dict(P0)           # sess.run(init) / saver.restore
    got = []
    for b in batches:
        cap_enc, cap_dec, cap_len = b["cap_enc"], b["cap_dec"], b["lengths"]
        if fine:                                                                   # main.py:74-81
            image_embeddings = vgg16(b["images"], trainable_fe=p2.fine_tune_fe, trainable_top=p2.fine_tune_top, dropout_keep=p2.cnn_dropout, params=p2)
            features = image_embeddings.fc2
        else:
            features = np.repeat(b["features"], nc, axis=0)                        # main.py:84-89: tiled x num_captions
        images_fv = layers.dense(features, p2.embed_size, name="imf_emb", params=p2)          # main.py:94
        encoder = None if p2.no_encoder else Encoder(images_fv, cap_enc, cap_len, p2)          # main.py:99-102
        decoder = Decoder(images_fv, cap_dec, cap_len, p2, CapDict)
        if p2.no_encoder:
            session.stage(p2, cap_enc=cap_enc)                                     # the labels of main.py:153 (no Encoder to carry them)
        if spec.uses_ci(p2):                                                       # main.py:104-116
            c_i_emb = layers.dense(b["c_v"], p2.embed_size, name="cv_emb", params=p2)
            decoder.c_i, decoder.c_i_ph = c_i_emb, b["c_v"]
            if encoder is not None:
                encoder.c_i, encoder.c_i_ph = c_i_emb, b["c_v"]
        obs = {}
        if encoder is not None:
            qz, tm_list, tv_list = encoder.q_net()                                 # main.py:117
            obs = {"z": qz}
        dec_model, x_logits, shpe, _ = decoder.px_z_fi(obs)                        # main.py:148-150
        assert tuple(x_logits.shape) == (B * nc * cap_dec.shape[1], V)
        lower_bound = tr.cap.fw_loss()                                             # main.py:152-177 (loss glue, engine.fw_loss)
        optimize, global_step, global_norm = optimizers.non_cnn_optimizer(lower_bound, p2)    # main.py:179-180
        optimize()
        if fine:
            optimize_cnn, _ = optimizers.cnn_optimizer(lower_bound, p2)            # main.py:181-183
            optimize_cnn()
        got.append(tr.losses())
    assert got == ref_losses, (got, ref_losses)
    assert got[0] != got[1]
    a, b2 = ref.state_dict(), tr.state_dict()
    for k in a:
        np.testing.assert_array_equal(a[k], b2[k], err_msg=k)


def test_facade_arrays_refilled_in_place_and_direct_set_batch_between(lib):
    """session.bind uploads what the facades hold at EVERY step: (1) a caller that refills preallocated arrays in place trains on
    the new contents (object identity says nothing), (2) a direct Trainer.set_batch between two facade steps (a validation batch)
    does not leave that batch resident for the next facade step, (3) a facade rebuilt with None forgets the arrays IT staged earlier (owner-scoped: never another facade's)."""
    from vae_captioning_amd import layers
    rng = np.random.default_rng(9)
    p1, p2 = _params(prior="Normal"), _params(prior="Normal")
    V = 90
    P0 = spec.init_caption_params(p1, V, seed=4)
    bs = [synth.make_batch(rng, 3, 2, 6, V, variable_len=True, feature_size=48) for _ in range(3)]
    ref = Trainer(p1, V, lib=lib, seed=5)
    ref.load_state_dict(P0)
    want = []
    for b in (bs[0], bs[1], bs[0]):
        ref.set_batch(b)
        ref.train_step()
        want.append(ref.losses())
    tr = session.get(p2)
    tr.cap.seed = 5
    tr.load_state_dict(P0)
    hold = {k: np.array(bs[0][k]) for k in ("features", "cap_enc", "cap_dec", "lengths")}   # the caller's preallocated arrays
    images_fv = layers.dense(hold["features"], p2.embed_size, name="imf_emb", params=p2)
    enc = Encoder(images_fv, hold["cap_enc"], hold["lengths"], p2)
    dec = Decoder(images_fv, hold["cap_dec"], hold["lengths"], p2, None)
    optimize, _, _ = optimizers.non_cnn_optimizer(None, p2)

    def step():
        z, _, _ = enc.q_net()
        dec.px_z_fi({"z": z})
        tr.cap.fw_loss()
        optimize()
        return tr.losses()
    got = [step()]
    for k in hold:                      # (1) refill in place: the same ndarray objects, other contents
        hold[k][...] = bs[1][k]
    got.append(step())
    tr.set_batch(bs[2])                 # (2) somebody uploads another batch directly ...
    for k in hold:
        hold[k][...] = bs[0][k]         # ... and the facades' arrays go back to the first contents
    got.append(step())
    assert got == want, (got, want)
    Encoder(None, None, None, p2)       # (3) a facade rebuilt with None forgets what IT staged -- not what the other facade staged
    assert "cap_enc" not in session.staged(p2) and "lengths" in session.staged(p2) and "cap_dec" in session.staged(p2)
    Decoder(None, None, None, p2, None)
    assert "cap_dec" not in session.staged(p2) and "lengths" not in session.staged(p2)


def test_facades_refuse_incomplete_or_embedded_inputs(lib):
    from vae_captioning_amd import layers
    p = _params(prior="Normal")
    rng = np.random.default_rng(5)
    b = synth.make_batch(rng, 3, 2, 6, 90, feature_size=48)
    with pytest.raises(TypeError):
        Encoder(np.zeros((6, 32), np.float32), b["cap_enc"], b["lengths"], p)      # an already embedded array
    images_fv = layers.dense(b["features"], p.embed_size, name="imf_emb", params=p)
    enc = Encoder(images_fv, b["cap_enc"], b["lengths"], p)
    with pytest.raises(ValueError, match="cap_dec"):
        enc.q_net()                                                                # no Decoder yet: the step's inputs are incomplete
    with pytest.raises(ValueError):
        layers.dense(b["features"], p.embed_size, name="fc9", params=p)


def test_make_rnn_cell_single_step_matches_oracle(lib):
    rng = np.random.default_rng(0)
    N, E, H = 5, 32, 64
    cell = make_rnn_cell([H], dropout_keep_prob=1.0)
    W = (rng.standard_normal((E + H, 4 * H)) * 0.2).astype(np.float32)
    b = (rng.standard_normal(4 * H) * 0.1).astype(np.float32)
    cell.bind(torch.from_numpy(W).cuda(), torch.from_numpy(b).cuda())
    x = rng.standard_normal((N, E)).astype(np.float32)
    state = rnn_placeholders(cell.zero_state(N))
    out, state = cell(torch.from_numpy(x).cuda(), state)
    out2, state2 = cell(torch.from_numpy(x).cuda(), state)
    c = O.lstm_seq_fwd(np.stack([x, x]).astype(np.float64), np.full(N, 2), W.astype(np.float64), b.astype(np.float64))
    np.testing.assert_allclose(out.cpu().numpy(), c["hs"][1], rtol=0, atol=2e-6)
    np.testing.assert_allclose(state2[0].c.cpu().numpy(), c["cs"][2], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out2.cpu().numpy(), c["hs"][2], rtol=0, atol=2e-6)
    with pytest.raises(NotImplementedError):
        make_rnn_cell([H, H])


def test_main_cli_training_then_inference(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = [sys.executable, os.path.join(ROOT, "main.py"), "--synthetic", "--vocab", "300", "--bs", "4", "--embed_dim", "32",
            "--enc_hid", "64", "--dec_hid", "64", "--latent", "10", "--gen_z_samples", "4", "--gpu", "0", "--checkpoint", "clitest"]
    r = subprocess.run(base + ["--epochs", "1", "--max_steps", "3", "--prior", "AG", "--c_v"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Validation reconstruction loss" in r.stdout and "Model saved in file" in r.stdout
    # saver.save writes TF V2 checkpoint files (main.py:286-288)
    from vae_captioning_amd import tf_bundle
    assert sorted(os.listdir(tmp_path / "checkpoints")) == ["checkpoint", "clitest.ckpt.data-00000-of-00001", "clitest.ckpt.index"]
    assert tf_bundle.latest_checkpoint(str(tmp_path / "checkpoints")) == str(tmp_path / "checkpoints" / "clitest.ckpt")
    z = tf_bundle.read_bundle(str(tmp_path / "checkpoints" / "clitest.ckpt"))
    assert "encoder/ag_ll_89/dense_1/kernel" in z and z["decoder/rnn_logits/kernel"].shape == (64, 300)
    assert z["decoder/net/z_rnn/kernel"].shape == (4 * 10, 32)
    r = subprocess.run(base + ["--mode", "inference", "--sample_gen", "greedy", "--prior", "AG", "--c_v", "--gen_name", "t1"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    caps = json.load(open(tmp_path / "val_t1.json"))
    assert len(caps) == 8 and set(caps[0]) == {"image_id", "caption"}
    # --fine_tune on synthetic images (VGG16 on device, CNN optimiser, npz checkpoint format)
    r = subprocess.run(base + ["--epochs", "1", "--max_steps", "2", "--fine_tune", "--bs", "2", "--ckpt_format", "npz", "--checkpoint", "ft"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(tmp_path / "checkpoints" / "ft.ckpt.npz")
    assert z["cnn/fc1/weights"].shape == (25088, 4096) and "decoder/rnn_logits/kernel" in z.files


def test_bench_two_ranks_through_torchrun_on_one_gpu(tmp_path):
    """The driver launches bench.py for N > 1 as `python -m torch.distributed.run --nproc-per-node N ...`.
    Only one GPU is available to the tests, so two ranks share it over gloo (VC_DIST_BACKEND); this
    exercises rank/env handling, the count / mean-std / gradient collectives, the max-over-ranks
    timing and the single JSON line on rank 0."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PYTHONPATH=ROOT, VC_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "cfg1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["global_caption_rows"] == 2 * d["config"]["caption_rows_per_gpu"]
    assert d["value"] > 0 and d["scaling"] == "weak" and d["roofline"]["achieved"] > 0
    cmd[cmd.index("cfg1")] = "cfg2"  # encoder present: exercises the all-gather / reduce-scatter of the global Q1 mix
    r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and np.isfinite(d["final_losses"]["rec_loss"]) and np.isfinite(d["final_losses"]["kld"])
    # the headline workload (VGG16 fine-tuning): four asynchronous gradient buckets under the three-stream backward
    cmd[cmd.index("cfg2")] = "cfg4"
    r = subprocess.run(cmd + ["--images-per-gpu", "2"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["global_caption_rows"] == 20 and np.isfinite(d["final_losses"]["rec_loss"])
    # the workload description survives the data-parallel statistics (a local of the bucket-wait summary once shadowed it)
    assert d["config"]["workload"].startswith("cfg4: {") and '"fine_tune": true' in d["config"]["workload"]
    assert len(d["data_parallel"]["bucket_wait_ms"]) == 4 and d["data_parallel"]["rccl_world_size"] == 2
    assert d["data_parallel"]["collectives"].startswith("torch.distributed") and d["data_parallel"]["compute_only_ms_per_step"] > 0
    assert "exposed_comm_ms" in d["data_parallel"]
    # VC_DP_COMM=abi asks for libvaecap's own RCCL communicator: RCCL refuses two ranks on ONE GPU (vc_comm_init_rank -> non-zero code with
    # RCCL's message on both ranks), the ranks agree on that and fall back to torch.distributed together -- same JSON line, a notice on rank 0
    cmd[cmd.index("cfg4")] = "cfg2"
    r = subprocess.run(cmd, cwd=tmp_path, env=dict(env, VC_DP_COMM="abi"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "libvaecap communicator not available on every rank" in r.stdout and "ncclCommInitRank failed" in r.stdout
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and np.isfinite(d["final_losses"]["rec_loss"]) and d["data_parallel"]["collectives"].startswith("torch.distributed")


def test_bench_gpus_2_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2 ...` with no torchrun wrapper and no WORLD_SIZE: bench.py starts its two ranks itself
    (bench.self_launch), rank 0 prints the ONE JSON line, exit code 0.  Two ranks share the one GPU over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT, VC_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg4",
           "--images-per-gpu", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["config"]["global_caption_rows"] == 20
    assert d["value"] > 0 and np.isfinite(d["final_losses"]["rec_loss"])
    # a launcher whose world size disagrees with --gpus is an error message, not a traceback
    r = subprocess.run(cmd, cwd=tmp_path, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr


def test_gen_caption_cli_single_image(tmp_path, lib):
    """gen_caption.py (:134-165): params pickle + vocabulary pickle + TF-format checkpoint + an image file ->
    caption.  LSTM baseline (--no_encoder: deterministic); the token ids must equal the oracle's greedy decode
    of the features the device VGG16 extracts from the same resized image."""
    import pickle
    from PIL import Image
    from oracle import decode as od
    from vae_captioning_amd import tf_bundle
    from vae_captioning_amd.trainer import VggEngine
    from vae_captioning_amd.utils.captions import Dictionary
    from vae_captioning_amd.utils.image_utils import keras_load_img
    rng = np.random.default_rng(21)
    words = ["w%02d" % i for i in range(25)]
    caps = {"img%d.jpg" % i: [["<BOS>"] + [words[j] for j in rng.integers(0, 25, size=6)] + ["<EOS>"] for _ in range(5)] for i in range(12)}
    d = Dictionary(caps, 3)
    V = d.vocab_size
    p = Parameters()
    p.embed_size, p.decoder_hidden, p.encoder_hidden, p.latent_size, p.gen_z_samples = 32, 64, 64, 10, 4
    p.no_encoder, p.gen_max_len = True, 9
    PC = spec.init_caption_params(p, V, seed=3)
    for k in PC:
        PC[k] = (PC[k] * 3).astype(np.float32)
    PV = spec.init_vgg_params(seed=4)
    os.makedirs(tmp_path / "checkpoints")
    os.makedirs(tmp_path / "pickles")
    ck = str(tmp_path / "checkpoints" / "gc.ckpt")
    tf_bundle.write_bundle(ck, {**PC, **PV})
    with open(tmp_path / "pickles" / "params.pickle", "wb") as f:
        pickle.dump(dict(embed_size=32, decoder_hidden=64, encoder_hidden=64, latent_size=10, gen_z_samples=4, no_encoder=True,
                         gen_max_len=9, keep_words=3, use_c_v=False, prior="Normal"), f)
    with open(tmp_path / "pickles" / "capt_vocab.pickle", "wb") as f:
        pickle.dump(caps, f)
    img = rng.integers(0, 256, size=(200, 300, 3), dtype=np.uint8)
    Image.fromarray(img).save(tmp_path / "pic.png")
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "gen_caption.py"), "--img_path", str(tmp_path / "pic.png"), "--checkpoint", ck,
           "--params_path", str(tmp_path / "pickles" / "params.pickle"), "--vocab_path", str(tmp_path / "pickles" / "capt_vocab.pickle"),
           "--gpu", "0"]
    r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    caption = r.stdout.strip().splitlines()[-1]
    # expected: device features of the same NEAREST-resized image -> oracle greedy decode
    x, _ = keras_load_img(str(tmp_path / "pic.png"))
    assert x.shape == (1, 224, 224, 3) and x.dtype == np.float32
    pv = Parameters()
    pv.mode = "inference"
    vgg = VggEngine(pv, lib=lib)
    vgg.load_params(PV)
    feats = vgg.forward(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
    P64 = {k: v.astype(np.float64) for k, v in PC.items()}
    ref = od.greedy(P64, p, feats[0], None, None, d.word2idx["<BOS>"], d.word2idx["<EOS>"], c_means=None, max_len=9)
    want = " ".join(d.idx2word[t] for t in ref if t not in (d.word2idx["<BOS>"], d.word2idx["<EOS>"]))
    assert len(ref) > 0 and caption == want, (caption, want)
    r = subprocess.run(cmd + ["--gen_method", "beam_search", "--beam_size", "3"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and len(r.stdout.strip().splitlines()[-1]) > 0, r.stdout + r.stderr


def test_coco_directory_training_and_inference(tmp_path, lib):
    """The reference's own data path (main.py:23-41, utils/data.py, ops/inference.py) on a miniature MSCOCO tree:
    batched VGG16 feature extraction into ./pickles/*.pickle, training from the generators, val / test json."""
    import pickle
    from . import coco_fixture
    from vae_captioning_amd.trainer import VggEngine
    from vae_captioning_amd.utils.image_utils import load_image
    coco = coco_fixture.build(tmp_path / "coco")
    os.makedirs(tmp_path / "utils")
    PV = coco_fixture.vgg_weight_file(str(tmp_path / "utils" / "vgg16_weights.npz"))
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = [sys.executable, os.path.join(ROOT, "main.py"), "--coco_dir", coco, "--bs", "2", "--embed_dim", "32", "--enc_hid", "64",
            "--dec_hid", "64", "--latent", "10", "--gen_z_samples", "4", "--gpu", "0", "--checkpoint", "cocotest"]
    r = subprocess.run(base + ["--epochs", "1", "--max_steps", "3"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Extracting features" in r.stdout and "Validation reconstruction loss" in r.stdout
    assert sorted(os.listdir(tmp_path / "pickles")) == ["capt_vocab.pickle", "test2014.pickle", "train2014.pickle", "val2014.pickle"]
    feats = pickle.load(open(tmp_path / "pickles" / "train2014.pickle", "rb"))
    assert len(feats) == 6 and all(v.shape == (1, 4096) and v.dtype == np.float32 for v in feats.values())
    # the pickled features are the device VGG16's fc2 of the cv2-style resized images, whatever the batching
    pv = Parameters()
    pv.mode = "inference"
    vgg = VggEngine(pv, lib=lib)
    vgg.load_params(PV)
    name = sorted(feats)[3]
    img = load_image(coco + "images/train2014/" + name).astype(np.float32)[None]
    one = vgg.forward(torch.from_numpy(img).cuda()).cpu().numpy()
    # (a batch of 6 and a batch of 1 cut their convolutions into different main / K-split tail launches: same values up to
    # fp32 summation order)
    assert np.linalg.norm(feats[name] - one) <= 1e-5 * np.linalg.norm(one)
    # second run loads the cached pickles; inference writes both json files with COCO image ids
    r = subprocess.run(base + ["--mode", "inference", "--sample_gen", "greedy", "--gen_name", "c1"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Loading prepared feature vector" in r.stdout and "Extracting features" not in r.stdout
    val = json.load(open(tmp_path / "val_c1.json"))
    test = json.load(open(tmp_path / "test_c1.json"))
    assert len(val) == 4 and len(test) == 2 and all(isinstance(c["image_id"], int) and isinstance(c["caption"], str) for c in val + test)
    # fine-tuning reads the images themselves (no preprocessed array here: files are decoded per batch) and starts
    # from the ImageNet weight file
    r = subprocess.run(base + ["--epochs", "1", "--max_steps", "2", "--fine_tune", "--checkpoint", "cocoft"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Loading imagenet weights" in r.stdout and "decoding the image files per batch" in r.stdout
    from vae_captioning_amd import tf_bundle
    z = tf_bundle.read_bundle(str(tmp_path / "checkpoints" / "cocoft.ckpt"), names=["cnn/conv1_1/weights", "cnn/fc2/biases"])
    assert z["cnn/conv1_1/weights"].shape == (3, 3, 3, 64) and not np.array_equal(z["cnn/conv1_1/weights"], PV["cnn/conv1_1/weights"])
