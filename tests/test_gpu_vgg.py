"""-m gpu: VGG16 engine (13 convs, 5 pools, fc1/fc2 + dropout, backward, CNN Adam with L2) and
the fine-tune training step through the C ABI vs the fp64 CPU oracle, B = 1 at 224x224.

Forward: every layer's activation vs the pure oracle, relative L2 <= 2e-5.
Backward: ReLU masks and pool arg-max are discontinuous, and a single decision that flips
between the fp32 device sum and the fp64 oracle sum moves a whole conv5 weight gradient by
~2e-3 relative (one of 25088 pool5 windows re-routed).  The backward pass is therefore
checked against the oracle's backward evaluated ON THE DEVICE'S forward activations (same
masks, same arg-max; the activations themselves are pinned by the forward check), relative
L2 <= 1e-4 per tensor; per-kernel max-abs parity is asserted in test_gpu_ops.py."""
import numpy as np
import pytest
import torch

from oracle import caption_model as cm
from oracle import optim as oo
from oracle import vgg as ov
from vae_captioning_amd import spec, synth
from vae_captioning_amd.trainer import Trainer, VggEngine
from vae_captioning_amd.utils.parameters import Parameters

pytestmark = pytest.mark.gpu


def nhwc(t):
    """a VggEngine activation -- C4 layout [B, C/4, H, W, 4] between conv1_1 and pool5 (include/vaecap.h) -- as an NHWC numpy array"""
    a = t.detach().cpu().numpy()
    if a.ndim == 5:
        B, C4, H, W, _ = a.shape
        a = a.transpose(0, 2, 3, 1, 4).reshape(B, H, W, C4 * 4)
    return a


def device_cache(eng, P64, keep):
    """oracle.vgg cache rebuilt from the device's forward activations (fp32 values, as fp64)."""
    f = lambda t: nhwc(t).astype(np.float64)
    conv = []
    for name, x, H, W, ci, co, w in eng.acts:
        x64 = f(x)
        if name == "P":
            _, arg = ov.maxpool_fwd(x64)
            conv.append(("P", x64.shape, arg))
        else:
            if ci == 4 and name == "conv1_1":
                x64 = x64[..., :3]
            conv.append((name, x64, f(eng.buf["y_" + name])))
    d1 = f(eng.buf["drop1"]) if keep < 1 else None
    d2 = f(eng.buf["drop2"]) if keep < 1 else None
    return dict(conv=conv, pool5_shape=tuple(eng.flat.shape), flat=f(eng.flat).reshape(eng.B, -1), fc1=f(eng.buf["fc1"]),
                fc1d=f(eng.fc1d), fc2=f(eng.buf["fc2"]), drop1=d1, drop2=d2, keep=keep)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def test_vgg_forward_backward(lib):
    p = Parameters()
    p.fine_tune = True
    rng = np.random.default_rng(2)
    PV = spec.init_vgg_params(seed=3)
    img = rng.integers(0, 256, size=(1, 224, 224, 3)).astype(np.float32)
    d1 = (rng.random((1, 4096)) < 0.5).astype(np.float32)
    d2 = (rng.random((1, 4096)) < 0.5).astype(np.float32)
    dfc2 = rng.normal(size=(1, 4096)).astype(np.float32)
    P64 = {k: v.astype(np.float64) for k, v in PV.items()}
    fc2_ref, cache = ov.forward(P64, img.astype(np.float64), d1.astype(np.float64), d2.astype(np.float64), keep=0.5)
    Gref = ov.backward(P64, cache, dfc2.astype(np.float64))
    eng = VggEngine(p, lib=lib)
    eng.load_params(PV)
    eng.set_masks(d1, d2)
    fc2 = eng.forward(torch.from_numpy(img).cuda())
    assert rel_l2(fc2.cpu().numpy(), fc2_ref) < 2e-5
    # intermediate activations, layer by layer
    convs = [c for c in cache["conv"] if c[0] != "P"]
    for name, x, y in convs:
        got = nhwc(eng.buf["y_" + name])
        assert rel_l2(got, y) < 2e-5, name
    eng.backward(torch.from_numpy(dfc2).cuda())
    G = eng.grads_dict()
    Gdev = ov.backward(P64, device_cache(eng, P64, 0.5), dfc2.astype(np.float64))
    worst = max(rel_l2(G[n], Gref[n]) for n in Gref)
    assert worst < 5e-2, ("end-to-end vs pure oracle (loose: decision flips)", worst)
    for n, ref in Gdev.items():
        assert rel_l2(G[n], ref) < 1e-4, (n, rel_l2(G[n], ref))


def test_fine_tune_step_matches_oracle(lib):
    p = Parameters()
    p.fine_tune = True
    p.num_captions = 2
    p.gen_z_samples = 4
    V, B, T = 300, 1, 5
    rng = np.random.default_rng(5)
    PC = spec.init_caption_params(p, V, seed=1)
    PV = spec.init_vgg_params(seed=3)
    batch = synth.make_batch(rng, B, p.num_captions, T, V, images=True, variable_len=True)
    noise = synth.make_noise(rng, B * p.num_captions, T, p)
    noise["cnn_drop1"] = (rng.random((B, 4096)) < 0.5).astype(np.float32)
    noise["cnn_drop2"] = (rng.random((B, 4096)) < 0.5).astype(np.float32)
    # oracle, fp64
    f64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
    PV64, PC64, b64, n64 = f64(PV), f64(PC), f64(batch), f64(noise)
    fc2_ref, _ = ov.forward(PV64, b64["images"], n64["cnn_drop1"], n64["cnn_drop2"], keep=p.cnn_dropout)
    reg = float(ov.l2_reg_loss(PV, p.weight_decay))
    # device
    tr = Trainer(p, V, lib=lib)
    tr.load_state_dict({**PC, **PV})
    tr.set_batch(batch, noise)
    tr.train_step()
    fc2_dev = (tr.vgg.buf["fc2d"] if tr.vgg.keep < 1 else tr.vgg.buf["fc2"]).cpu().numpy().astype(np.float64)
    assert rel_l2(fc2_dev, fc2_ref) < 2e-5
    # caption side and VGG backward of the oracle on the device's forward decisions (see module docstring)
    b64["features"] = fc2_dev
    out = cm.forward_backward(PC64, b64, n64, p, global_step=0, reg_loss=reg)
    GV = ov.backward(PV64, device_cache(tr.vgg, PV64, p.cnn_dropout), out.dfeatures)
    kld, rec, lb, ann = tr.losses()
    assert ann == 1.0  # main.py:163-164: annealing forced to 1 when fine-tuning
    assert abs(rec - float(out.rec_loss)) <= 2e-4 * abs(float(out.rec_loss)), (rec, float(out.rec_loss))
    assert abs(kld - float(out.kld)) <= 2e-4 * abs(float(out.kld)) + 1e-6
    assert reg > 0 and rec > reg
    Gc = tr.cap.grads_dict()
    for n, ref in out.grads.items():
        # fp32 BPTT vs fp64: 2.05e-4 observed on the decoder LSTM kernel (summation order of the K-split recurrence kernels)
        assert rel_l2(Gc[n], ref) < 3e-4, (n, rel_l2(Gc[n], ref))
    Gv = tr.vgg.grads_dict()
    for n, ref in GV.items():
        assert rel_l2(Gv[n], ref) < 2e-4, (n, rel_l2(Gv[n], ref))
    # CNN Adam with the L2 term: first step moves every weight by ~cnn_lr * sign(g + wd*w)
    st = {}
    PVn = {k: v.copy() for k, v in PV.items()}
    oo.adam_step(PVn, {k: v.astype(np.float32) for k, v in GV.items()}, st, p.cnn_lr, 1, l2=p.weight_decay)
    new = tr.vgg.state_dict()
    for n in ("cnn/conv3_2/weights", "cnn/fc2/weights", "cnn/conv1_1/biases"):
        upd = np.abs(PVn[n] - PV[n]).max()
        assert rel_l2(new[n] - PV[n], PVn[n] - PV[n]) < 2e-2, (n, rel_l2(new[n] - PV[n], PVn[n] - PV[n]), upd)


def test_cached_weight_norm_is_dropped_when_somebody_else_writes_the_parameters(lib):
    """The Adam update of the cnn/* variables leaves per-block sums of w^2 for the next step's regulariser term (no second pass over
    0.54 GB).  The cache must be USED on an undisturbed run, and dropped as soon as the parameters are written behind the
    optimiser's back (store.p.copy_, a view's in-place edit, load_state_dict): the next step's reported loss then carries the new
    weights' norm."""
    p = Parameters()
    p.fine_tune, p.num_captions, p.gen_z_samples = True, 2, 4
    V, B, T = 300, 1, 5
    rng = np.random.default_rng(6)
    batch = synth.make_batch(rng, B, p.num_captions, T, V, images=True, variable_len=True)
    tr = Trainer(p, V, lib=lib, seed=3)
    tr.load_state_dict({**spec.init_caption_params(p, V, seed=1), **spec.init_vgg_params(seed=3)})
    tr.set_batch(batch)
    tr.train_step()
    vg = tr.vgg
    assert vg.w2_valid and vg.store.p._version == vg._w2_version      # nothing in a step writes the parameters through torch
    tr.train_step()
    assert vg.w2_valid and vg.store.p._version == vg._w2_version
    red = torch.zeros(2, device="cuda")

    def both():   # (cached or recomputed by reg_sumsq, always recomputed here)
        vg.reg_sumsq(red.data_ptr())
        torch.cuda.synchronize()
        return float(red[0].item()), float((vg.store.p.double() ** 2).sum().item())
    got, want = both()
    assert abs(got - want) <= 1e-5 * want
    vg.store.param("cnn/fc1/weights").mul_(1.5)                        # an edit through a view of the flat buffer
    got, want2 = both()
    assert want2 > 1.2 * want and abs(got - want2) <= 1e-5 * want2, (got, want2, want)
    tr.train_step()                                                    # the next Adam update makes the cache valid again
    assert vg.w2_valid and vg.store.p._version == vg._w2_version
    vg.store.p.copy_(vg.store.p * 0.5)
    got, want3 = both()
    assert abs(got - want3) <= 1e-5 * want3 and want3 < 0.3 * want2


def test_fallback_kernels_without_the_winograd_path(lib, monkeypatch):
    """use_wino = False: every layer through the NHWC implicit-GEMM kernels of csrc/conv.hip behind layout conversions (conv1_1 on zero-padded
    4-channel weights, separate max-pool launches) -- the path taken for geometries the Winograd / conv1 kernels do not support, and an
    independent implementation of the whole VGG16: forward vs the fp64 oracle, backward vs the oracle on the device's forward decisions,
    and close to the default path."""
    p = Parameters()
    p.fine_tune = True
    rng = np.random.default_rng(21)
    PV = spec.init_vgg_params(seed=6)
    img = rng.integers(0, 256, size=(1, 224, 224, 3)).astype(np.float32)
    dfc2 = rng.normal(size=(1, 4096)).astype(np.float32)
    ones = np.ones((1, 4096), np.float32)
    P64 = {k: v.astype(np.float64) for k, v in PV.items()}
    fc2_ref, cache = ov.forward(P64, img.astype(np.float64), ones.astype(np.float64), ones.astype(np.float64), keep=0.5)
    res = {}
    for mode in ("0", "1"):
        eng = VggEngine(p, lib=lib)
        if mode == "0":   # (an attribute, not an environment switch: the product never takes this path on a VGG16 shape)
            eng.use_wino = False
        eng.load_params(PV)
        eng.set_masks(ones, ones)
        fc2 = eng.forward(torch.from_numpy(img).cuda())
        eng.backward(torch.from_numpy(dfc2).cuda())
        torch.cuda.synchronize()
        res[mode] = (fc2.clone(), eng.store.g.clone())
        if mode == "0":
            assert rel_l2(fc2.cpu().numpy(), fc2_ref) < 2e-5
            for name, x, y in [c for c in cache["conv"] if c[0] != "P"]:
                assert rel_l2(nhwc(eng.buf["y_" + name]), y) < 2e-5, name
            G = eng.grads_dict()
            Gdev = ov.backward(P64, device_cache(eng, P64, 0.5), dfc2.astype(np.float64))
            for n, ref in Gdev.items():
                assert rel_l2(G[n], ref) < 1e-4, (n, rel_l2(G[n], ref))
    assert (res["0"][0] - res["1"][0]).norm() <= 1e-5 * res["1"][0].norm()
    assert (res["0"][1] - res["1"][1]).norm() <= 2e-3 * res["1"][1].norm()


@pytest.mark.parametrize("env", [("VC_CONV_WINO", "0"), ("VC_VGG_STREAMS", "1")], ids=lambda e: "%s=%s" % e)
def test_alternative_convolution_paths_match_the_oracle(lib, monkeypatch, env):
    """The non-default convolution paths stay correct: VC_CONV_WINO=0 (the NHWC implicit-GEMM kernels behind layout conversions on every layer),
    VC_VGG_STREAMS=1 (serial schedule; B = 2 so that the default
    would have used half-batch chains and the ReLU-mask bits of both geometries are exercised): forward vs the fp64 oracle, backward vs
    the oracle on the device's forward decisions."""
    monkeypatch.setenv(*env)
    p = Parameters()
    p.fine_tune = True
    rng = np.random.default_rng(33)
    PV = spec.init_vgg_params(seed=8)
    B = 2
    img = rng.integers(0, 256, size=(B, 224, 224, 3)).astype(np.float32)
    dfc2 = rng.normal(size=(B, 4096)).astype(np.float32)
    ones = np.ones((B, 4096), np.float32)
    P64 = {k: v.astype(np.float64) for k, v in PV.items()}
    fc2_ref, cache = ov.forward(P64, img.astype(np.float64), ones.astype(np.float64), ones.astype(np.float64), keep=0.5)
    eng = VggEngine(p, lib=lib)
    assert eng.use_wino == (env != ("VC_CONV_WINO", "0"))
    eng.load_params(PV)
    eng.set_masks(ones, ones)
    for _ in range(2):   # the first step allocates
        fc2 = eng.forward(torch.from_numpy(img).cuda())
        eng.backward(torch.from_numpy(dfc2).cuda())
    torch.cuda.synchronize()
    assert rel_l2(fc2.cpu().numpy(), fc2_ref) < 2e-5
    for name, x, y in [c for c in cache["conv"] if c[0] != "P"]:
        assert rel_l2(nhwc(eng.buf["y_" + name]), y) < 2e-5, name
    G = eng.grads_dict()
    Gdev = ov.backward(P64, device_cache(eng, P64, 0.5), dfc2.astype(np.float64))
    for n, ref in Gdev.items():
        assert rel_l2(G[n], ref) < 1e-4, (n, rel_l2(G[n], ref))


def test_half_batch_chains_on_three_streams(lib, monkeypatch):
    """B = 2: the default three-stream schedule (two half-batch conv/pool chains + weight gradients on a
    third stream, trainer.VggEngine) vs the fp64 oracle forward, vs the oracle backward on the device's
    forward decisions, and equal to the serial one-stream schedule up to fp32 summation order (full- and half-batch
    launches cut their convolutions into different main / K-split tail launches); two steps: the first allocates."""
    p = Parameters()
    p.fine_tune = True
    rng = np.random.default_rng(12)
    PV = spec.init_vgg_params(seed=4)
    B = 2
    img = rng.integers(0, 256, size=(B, 224, 224, 3)).astype(np.float32)
    d1 = (rng.random((B, 4096)) < 0.5).astype(np.float32)
    d2 = (rng.random((B, 4096)) < 0.5).astype(np.float32)
    dfc2 = rng.normal(size=(B, 4096)).astype(np.float32)
    P64 = {k: v.astype(np.float64) for k, v in PV.items()}
    fc2_ref, cache = ov.forward(P64, img.astype(np.float64), d1.astype(np.float64), d2.astype(np.float64), keep=0.5)
    res = {}
    for ns in ("3", "1"):
        monkeypatch.setenv("VC_VGG_STREAMS", ns)
        eng = VggEngine(p, lib=lib)
        assert (eng.side2 is not None) == (ns == "3")
        eng.load_params(PV)
        eng.set_masks(d1, d2)
        for _ in range(2):
            fc2 = eng.forward(torch.from_numpy(img).cuda())
            eng.backward(torch.from_numpy(dfc2).cuda())
        torch.cuda.synchronize()
        res[ns] = (fc2.clone(), eng.store.g.clone())
        if ns == "3":
            assert rel_l2(fc2.cpu().numpy(), fc2_ref) < 2e-5
            for name, x, y in [c for c in cache["conv"] if c[0] != "P"]:
                assert rel_l2(nhwc(eng.buf["y_" + name]), y) < 2e-5, name
            G = eng.grads_dict()
            Gdev = ov.backward(P64, device_cache(eng, P64, 0.5), dfc2.astype(np.float64))
            for n, ref in Gdev.items():
                assert rel_l2(G[n], ref) < 1e-4, (n, rel_l2(G[n], ref))
    # (not bit-identical: the rows that fall into a K-split tail launch differ between full- and half-batch launches)
    assert (res["3"][0] - res["1"][0]).norm() <= 1e-5 * res["1"][0].norm()
    # gradients: a ReLU / arg-max decision that flips on a last-bit difference re-routes a whole window (module docstring)
    assert (res["3"][1] - res["1"][1]).norm() <= 2e-3 * res["1"][1].norm()


def test_load_weights_assigns_a_shuffled_npz_in_sorted_key_order(lib, tmp_path):
    """VggEngine.load_weights (utils/image_embeddings.py:240-246, quirk Q18) at the real VGG16 shapes: the archive stores its
    arrays in shuffled order and carries fc8_*; every cnn/* variable must receive the array of the same rank in SORTED key
    order (conv1_1_W, conv1_1_b, ... fc6_W, fc6_b, fc7_W, fc7_b) and fc8 must be ignored."""
    keys = ["conv%d_%d_%s" % (b, i, sfx) for b, reps in ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3)) for i in range(1, reps + 1) for sfx in ("W", "b")]
    keys += ["fc6_W", "fc6_b", "fc7_W", "fc7_b"]
    shapes = [s for _, s in spec.vgg_variables()]
    arrays = {k: np.full(shp, 0.001 * (i + 1), np.float32) for i, (k, shp) in enumerate(zip(keys, shapes))}
    arrays["fc8_W"], arrays["fc8_b"] = np.full((4096, 1000), 7.0, np.float32), np.full((1000,), 7.0, np.float32)
    order = list(arrays)
    np.random.default_rng(1).shuffle(order)
    path = str(tmp_path / "vgg16_weights.npz")
    np.savez(path, **{k: arrays[k] for k in order})
    p = Parameters()
    p.fine_tune = True
    vgg = VggEngine(p, lib=lib)
    vgg.load_weights(path)
    sd = vgg.state_dict()
    for i, (name, shp) in enumerate(spec.vgg_variables()):
        assert sd[name].shape == tuple(shp) and np.all(sd[name] == np.float32(0.001 * (i + 1))), name


def test_checkpoints_carry_cnn_variables_without_fine_tune_and_restore_into_fine_tune(lib, tmp_path):
    """main.py:186-191 + quirk Q22: the reference's checkpoints always hold cnn/* (ImageNet weights are loaded "for further
    usage" even when training on precomputed features).  A checkpoint written WITHOUT --fine_tune must therefore restore
    into a --fine_tune model; one that lacks cnn/* falls back to the ImageNet npz or fails with a clear message."""
    from vae_captioning_amd.trainer import imagenet_weights
    p = Parameters()
    p.embed_size, p.encoder_hidden, p.decoder_hidden, p.latent_size, p.gen_z_samples = 16, 32, 32, 6, 3
    V = 50
    P0 = spec.init_caption_params(p, V, seed=1)
    PV = spec.init_vgg_params(seed=2)
    tr = Trainer(p, V, lib=lib)
    tr.load_state_dict(P0)
    assert not any(k.startswith("cnn/") for k in tr.state_dict())
    tr.cnn_host = PV                                    # what main.py sets from the ImageNet npz
    ck = str(tmp_path / "plain.ckpt.npz")
    tr.save(ck)
    p2 = Parameters()
    for k in ("embed_size", "encoder_hidden", "decoder_hidden", "latent_size", "gen_z_samples"):
        setattr(p2, k, getattr(p, k))
    p2.fine_tune = True
    tr2 = Trainer(p2, V, lib=lib)
    tr2.restore(ck)
    got = tr2.state_dict()
    for k in list(PV)[:4] + list(PV)[-2:]:
        np.testing.assert_array_equal(got[k], PV[k])
    # restoring a checkpoint that has cnn/* into a model without VGG16 keeps them for the next save
    tr3 = Trainer(p, V, lib=lib)
    tr3.restore(ck)
    assert set(k for k in tr3.state_dict() if k.startswith("cnn/")) == set(PV)
    # a checkpoint without cnn/*: clear error, or the ImageNet file when it exists
    ck2 = str(tmp_path / "nocnn.ckpt.npz")
    np.savez(ck2, **P0)
    p2.image_net_weights_path = str(tmp_path / "missing.npz")
    with pytest.raises(KeyError, match="cnn/"):
        Trainer(p2, V, lib=lib).restore(ck2)
    inet = str(tmp_path / "vgg16_weights.npz")
    keys = ["conv%d_%d_%s" % (b, i, sfx) for b, reps in ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3)) for i in range(1, reps + 1) for sfx in ("W", "b")]
    keys += ["fc6_W", "fc6_b", "fc7_W", "fc7_b"]
    np.savez(inet, **{k: PV[n] for k, (n, _) in zip(keys, spec.vgg_variables())})
    p2.image_net_weights_path = inet
    tr4 = Trainer(p2, V, lib=lib)
    tr4.restore(ck2)
    np.testing.assert_array_equal(tr4.state_dict()["cnn/fc2/biases"], imagenet_weights(inet)["cnn/fc2/biases"])


def test_uint8_images_train_bit_identically_to_the_float32_feed(lib):
    """The reference's HDF5 file holds uint8 pixels (preprocess.py:27-28) and feeds them into a float32 placeholder
    (utils/image_embeddings.py:31-34).  Trainer.set_batch ships a uint8 array as bytes (a quarter of the copy) and casts on the device
    (vc_vgg_preprocess_u8): the step must be bit-identical to feeding the same pixels as float32."""
    p = Parameters()
    p.fine_tune, p.num_captions, p.gen_z_samples = True, 2, 4
    V, B, T = 300, 2, 5
    rng = np.random.default_rng(6)
    P0 = {**spec.init_caption_params(p, V, seed=1), **spec.init_vgg_params(seed=3)}
    batch = synth.make_batch(rng, B, p.num_captions, T, V, images="u8")
    assert batch["images"].dtype == np.uint8
    res = []
    for cast in (False, True):
        b = dict(batch, images=batch["images"].astype(np.float32)) if cast else batch
        tr = Trainer(p, V, lib=lib, seed=5)
        tr.load_state_dict(P0)
        tr.set_batch(b)
        assert tr.images.dtype == (torch.float32 if cast else torch.uint8)
        tr.train_step()
        torch.cuda.synchronize()
        res.append((tr.losses(), tr.gall.clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
