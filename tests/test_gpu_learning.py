"""-m gpu: does the trainer LEARN, and does it stay on the oracle's trajectory beyond the 3-step golden pins?

(a) Behaviour (the purpose of main.py:213-260): Adam steps on a fixed batch of sixteen captions until the reconstruction loss has
    fallen below 0.1, then greedy decoding (vae_model/decoder.py:145-201) must return every training caption token for token -- the
    whole loop forward -> loss -> gradients -> clip -> Adam -> generation, not one step against a checker.  f32 and split-bf16.
(b) Trajectory: 50 CONSECUTIVE optimiser steps against the fp64 oracle stepping the same parameters with the oracle's own Adam
    (oracle/optim.py): loss / KL within the north-star 1e-3 at EVERY step in f32; the split-bf16 drift is measured and bounded.
Independent of tests/golden/step_*.npz (their three steps are regression pins of the oracle)."""
import numpy as np
import pytest

from oracle import caption_model as cm
from oracle import decode, optim as oo
from vae_captioning_amd import spec, synth
from vae_captioning_amd.generate import CaptionGenerator
from vae_captioning_amd.trainer import Trainer
from vae_captioning_amd.utils.parameters import Parameters

pytestmark = pytest.mark.gpu


def _params(**kw):
    p = Parameters()
    p.embed_size, p.encoder_hidden, p.decoder_hidden = 64, 128, 128
    p.latent_size, p.gen_z_samples, p.cnn_feature_size = 20, 6, 96
    for k, v in kw.items():
        setattr(p, k, v)
    return p


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("kw", [dict(no_encoder=True), dict(prior="Normal")], ids=["lstm-baseline", "normal-cvae"])
def test_trainer_memorises_sixteen_captions_and_greedy_decoding_returns_them(lib, kw, precision):
    p = _params(num_captions=1, batch_size=16, learning_rate=4e-3, **kw)
    V, B, T, STEPS = 200, 16, 9, 150
    rng = np.random.default_rng(42)
    batch = synth.make_batch(rng, B, 1, T, V, variable_len=True, feature_size=p.cnn_feature_size)
    assert batch["lengths"].min() >= 6   # (no all-PAD rows in this draw: sixteen real captions of 6..9 tokens)
    tr = Trainer(p, V, lib=lib, precision=precision, seed=5)
    tr.load_state_dict(spec.init_caption_params(p, V, seed=3))
    tr.set_batch(batch)   # noise (eps) drawn on device, a fresh draw every step: what a training run does
    first = None
    for s in range(STEPS):
        tr.train_step()
        if s == 0:
            first = tr.losses()[1]
    kld, rec, lb, ann = tr.losses()
    assert first > 4.0 and rec < 0.1, (first, rec)   # ln(200) = 5.3 at the start
    gen = CaptionGenerator(tr.cap)
    eps = None if p.no_encoder else rng.standard_normal((p.gen_z_samples, B, p.latent_size)).astype(np.float32)
    got = gen.greedy(batch["features"], None, eps, synth.BOS, synth.EOS, max_len=T + 3)
    for b in range(B):
        l = int(batch["lengths"][b])
        assert got[b] == batch["cap_enc"][b, :l].tolist(), (b, got[b], batch["cap_enc"][b, :l].tolist())
    if precision == "f32":
        # beam search on the TRAINED weights: identical beams to the fp64 oracle's (the reference's beam search feeds <BOS> twice,
        # vae_model/decoder.py:230-262, so its output is not the memorised caption -- a quirk kept bug for bug, SURVEY.md Q-list)
        P64 = {k: v.astype(np.float64) for k, v in tr.state_dict().items()}
        beams = gen.beam_search(batch["features"], None, eps, synth.BOS, synth.EOS, beam_size=3, max_len=T + 3)
        for b in range(4):
            sents, scores = decode.beam_search(P64, p, batch["features"][b].astype(np.float64), None, None if eps is None else eps[:, b:b + 1].astype(np.float64),
                                               synth.BOS, synth.EOS, beam_size=3, max_len=T + 3)
            assert [s_ for s_, _ in beams[b]] == sents, (b, beams[b], sents)


def test_a_generator_used_before_training_decodes_with_the_trained_weights(lib):
    """The generator keeps operands DERIVED from the parameters (decoder Wh in the step kernel's order, the vocabulary's input
    projections) across calls, keyed by CaptionEngine.param_version.  Every way the parameters change must invalidate them: optimiser
    steps (library kernels: torch's version counter does not move), load_state_dict and an in-place write through a view (it does)."""
    p = _params(num_captions=1, batch_size=8, learning_rate=4e-3, prior="Normal")
    V, B, T = 60, 8, 7
    rng = np.random.default_rng(7)
    batch = synth.make_batch(rng, B, 1, T, V, variable_len=True, feature_size=p.cnn_feature_size)
    tr = Trainer(p, V, lib=lib, seed=5)
    tr.load_state_dict(spec.init_caption_params(p, V, seed=3))
    tr.set_batch(batch)
    eps = rng.standard_normal((p.gen_z_samples, B, p.latent_size)).astype(np.float32)
    old = CaptionGenerator(tr.cap)
    decode_all = lambda g: (g.greedy(batch["features"], None, eps, synth.BOS, synth.EOS, max_len=T + 3),
                            g.beam_search(batch["features"], None, eps, synth.BOS, synth.EOS, beam_size=3, max_len=T + 3))
    before = decode_all(old)
    v0 = tr.cap.param_version
    assert decode_all(old) == before and tr.cap.param_version == v0      # (nothing changed: same version, cached operands reused)
    for _ in range(40):
        tr.train_step()
    assert tr.cap.param_version != v0
    after = decode_all(old)
    assert after == decode_all(CaptionGenerator(tr.cap)) and after != before
    v1 = tr.cap.param_version
    tr.cap.store.param("decoder/net/dec_embeddings").mul_(-1.0)           # a write through a view
    assert tr.cap.param_version != v1
    flipped = decode_all(old)
    assert flipped == decode_all(CaptionGenerator(tr.cap)) and flipped != after
    tr.load_state_dict(spec.init_caption_params(p, V, seed=3))             # back to the start
    assert decode_all(old) == before


@pytest.mark.parametrize("name,lr,kw", [("normal", 5e-4, dict(prior="Normal")), ("ag_cv", 2e-4, dict(prior="AG", use_c_v=True))])
def test_fifty_consecutive_steps_stay_on_the_oracles_trajectory(lib, name, lr, kw):
    """AG runs at lr = 2e-4: at 5e-4 its KL term (482 at the start, Q3's per-row sum) reaches its floor after ~40 steps, and from there
    Adam's sign-like steps oscillate round the minimum -- device and oracle agree to 1e-5 for forty steps and then part by 0.1 in KL
    within five (measured, both precisions): a property of the optimisation, not of the kernels; the slower rate keeps all fifty
    steps in the regime where a trajectory comparison means something."""
    p = _params(num_captions=2, batch_size=4, learning_rate=lr, **kw)
    V, B, T, STEPS, CLIP = 120, 4, 7, 50, 5.0
    rng = np.random.default_rng(7)
    P0 = spec.init_caption_params(p, V, seed=11)
    batch = synth.make_batch(rng, B, p.num_captions, T, V, use_ci=spec.uses_ci(p), variable_len=True, feature_size=p.cnn_feature_size)
    noise = synth.make_noise(rng, B * p.num_captions, T, p)
    f64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
    P64, b64, n64 = f64(P0), f64(batch), f64(noise)
    if p.prior == "AG":
        n64["c_means"] = decode.init_clusters(90, p.latent_size).astype(np.float64)
    ref, st = [], {}
    for s in range(STEPS):   # the oracle's own trajectory: fp64 forward / backward, float32 Adam (TF keeps float32 variables and slots)
        r = cm.forward_backward(P64, b64, n64, p, global_step=s)
        norm = float(oo.global_norm({k: v.astype(np.float32) for k, v in r.grads.items()}, {k: v.astype(np.float32) for k, v in r.sparse.items()}))
        ref.append((float(np.mean(r.kld)), float(r.rec_loss)))
        P32 = {k: v.astype(np.float32) for k, v in P64.items()}
        oo.adam_step(P32, {k: v.astype(np.float32) for k, v in r.grads.items()}, st, p.learning_rate, s + 1, scale=CLIP * min(1 / norm, 1 / CLIP))
        P64 = f64(P32)
    assert ref[-1][1] < ref[0][1] - 0.05   # the trajectory goes somewhere: the loss falls over the fifty steps
    drift = {}
    for precision in ("f32", "bf16x3"):
        tr = Trainer(p, V, lib=lib, precision=precision)
        tr.load_state_dict(P0)
        worst = 0.0
        for s in range(STEPS):
            tr.set_batch(batch, noise)
            tr.train_step()
            kld, rec, lb, ann = tr.losses()
            worst = max(worst, abs(rec - ref[s][1]), abs(kld - ref[s][0]) / max(1.0, abs(ref[s][0])))
            if s % 7 == 0 or s == STEPS - 1:
                print("  %s %s step %2d: rec %.6f (oracle %.6f)  kld %.6f (oracle %.6f)" % (name, precision, s, rec, ref[s][1], kld, ref[s][0]))
        drift[precision] = worst
        Q = tr.state_dict()
        moved = max(np.abs(P64[k] - P0[k]).max() for k in P0)
        far = max(np.abs(Q[k] - P64[k]).max() for k in P0)
        assert far <= (0.05 if precision == "f32" else 0.2) * moved, (precision, far, moved)   # parameters after 50 steps: within 5 % (20 %) of the distance travelled
    print("%s: worst |loss - oracle| over %d steps: f32 %.2e, bf16x3 %.2e" % (name, STEPS, drift["f32"], drift["bf16x3"]))
    assert drift["f32"] <= 1e-3, drift          # north_star: per-step loss / KL within 1e-3 in fp32
    assert drift["bf16x3"] <= 2e-4, drift       # the opt-in mode: measured 1.3e-5 / 2.7e-5, the same as f32 (the loss is summed in f32 either way)
