"""-m gpu: the split-bf16 ("bf16x3") products -- vc_gemm_bf16x3_f32 == vc_gemm_f32 with VC_GEMM_BF16X3, VC_LSTM_BF16X3 on the
sequence calls, Trainer(precision="bf16x3") on a whole step.  The mode is carried PER CALL (ABI 4): no test here restores anything.

NOT the reference's arithmetic (tf.float32 matmul): an opt-in mode reported on its own bench lines.  What it is held to:
  * every storage-order / tile-plan / split-K / flag combination the f32 kernel is tested on, against the fp64 product, with the
    tolerance the F(4x4,3x3) convolution path passes (6e-5 of the tensor maximum, flat) -- measured ~4e-6;
  * the error model: |C - ref| <= 2^-16 * (|A| . |B|) element by element (three bf16 x bf16 products exact in f32, the dropped
    lo.lo term and the two-term operand representation each ~2^-18 of |a b|, f32 accumulation on top);
  * a whole training step in this mode against the fp64 oracle: loss / KL within the north-star 1e-3, gradients within 1e-3 of
    their maximum."""
import numpy as np
import pytest
import torch

from .gpu_util import P, assert_close, dev, empty_bytes, host, stream, zeros

pytestmark = pytest.mark.gpu
TOL = 6e-5


def _run(lib, ta, tb, M, N, K, A, B, bias=None, flags=0, C0=None, ldc=None):
    ldc = ldc or N
    C = dev(C0) if C0 is not None else zeros(M, ldc)
    ws = empty_bytes(lib.vc_gemm_workspace_bytes(M, N, K))
    lib.vc_gemm_bf16x3_f32(stream(), ta, tb, M, N, K, P(dev(A.T if ta else A)), M if ta else K, P(dev(B.T if tb else B)), K if tb else N,
                           P(C), ldc, P(dev(bias)) if bias is not None else None, flags, P(ws), ws.numel() * 4)
    return host(C)


@pytest.mark.parametrize("ta", [0, 1])
@pytest.mark.parametrize("tb", [0, 1])
@pytest.mark.parametrize("shape", [(128, 128, 32), (256, 384, 64), (130, 70, 33), (64, 512, 1024), (37, 300, 700), (1000, 520, 96), (6, 40, 12), (512, 1000, 4096)],
                         ids=lambda s: "x".join(map(str, s)))
def test_bf16x3_gemm_matches_fp64(lib, ta, tb, shape):
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N * 3 + K + ta * 2 + tb)
    A = rng.standard_normal((M, K), dtype=np.float32)
    B = rng.standard_normal((K, N), dtype=np.float32)
    bias = rng.standard_normal(N, dtype=np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64) + bias
    got = _run(lib, ta, tb, M, N, K, A, B, bias)
    assert_close(got, ref, TOL, msg="bf16x3 gemm ta=%d tb=%d %s" % (ta, tb, shape))
    # element-wise error model (asymmetric operands: a transposed or mis-mapped fragment fails this by orders of magnitude)
    bound = 2.0 ** -16 * (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)) + 1e-6
    assert (np.abs(got - ref) <= bound).all(), float((np.abs(got - ref) / bound).max())


@pytest.mark.parametrize("ta,flags,shape", [(0, 0, (4224, 2052, 96)), (1, 3, (4100, 2052, 100)), (0, 1, (6880, 10000, 512)), (1, 2, (512, 10000, 6880)),
                                            (0, 3, (1000, 2048, 2560)), (1, 0, (256, 2048, 6880)), (0, 0, (64, 4096, 6272))],
                         ids=["unsplit-ragged-N", "KM-ragged-M-K", "logits-fwd", "logits-wgrad-splitK", "splitK-flags", "dWx-splitK", "skinny-fc"])
def test_bf16x3_gemm_training_shapes(lib, ta, flags, shape):
    M, N, K = shape
    rng = np.random.default_rng(M + N + K + flags)
    A = rng.standard_normal((M, K), dtype=np.float32)
    B = rng.standard_normal((K, N), dtype=np.float32)
    bias = rng.standard_normal(N, dtype=np.float32)
    C0 = rng.standard_normal((M, N + 4), dtype=np.float32)
    ref = C0.astype(np.float64).copy()
    prod = A.astype(np.float64) @ B.astype(np.float64) + bias
    if flags & 2:
        prod = prod + C0[:, :N]
    if flags & 1:
        prod = np.maximum(prod, 0)
    ref[:, :N] = prod
    got = _run(lib, ta, 0, M, N, K, A, B, bias, flags, C0, ldc=N + 4)
    assert_close(got, ref, TOL, msg="bf16x3 gemm ta=%d flags=%d %s (padding column untouched)" % (ta, flags, shape))


def test_bf16x3_is_much_closer_than_plain_bf16_and_handles_wide_dynamic_range(lib):
    """Operands spanning 2^-40 .. 2^40 per row (bf16 keeps f32's exponent range: no scaling needed) and the comparison that shows the
    lo terms are really there: a single-bf16 product is ~2^-9 off, the split one ~2^-17."""
    M, N, K = 256, 256, 512
    rng = np.random.default_rng(3)
    A = (rng.standard_normal((M, K)) * np.exp2(rng.integers(-40, 40, size=(M, 1)))).astype(np.float32)
    B = (rng.standard_normal((K, N)) * np.exp2(rng.integers(-40, 40, size=(1, N)))).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    got = _run(lib, 0, 0, M, N, K, A, B)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    rel = np.abs(got - ref) / scale
    assert rel.max() <= 2.0 ** -16, rel.max()
    tb = lambda x: (torch.from_numpy(x).to(torch.bfloat16).to(torch.float64)).numpy()
    plain = np.abs(tb(A) @ tb(B) - ref) / scale
    assert np.median(rel) * 50 < np.median(plain)


def test_precision_is_a_flag_of_the_call_and_the_deprecated_global_is_only_a_default(lib):
    M, N, K = 384, 256, 512
    rng = np.random.default_rng(8)
    A, B = rng.standard_normal((M, K), dtype=np.float32), rng.standard_normal((K, N), dtype=np.float32)
    dA, dB = dev(A), dev(B)
    ws = empty_bytes(lib.vc_gemm_workspace_bytes(M, N, K))

    def run(fn, flags):
        C = zeros(M, N)
        fn(stream(), 0, 0, M, N, K, P(dA), K, P(dB), N, P(C), N, None, flags, P(ws), ws.numel() * 4)
        return host(C).copy()
    f32, flag, entry = run(lib.vc_gemm_f32, 0), run(lib.vc_gemm_f32, 4), run(lib.vc_gemm_bf16x3_f32, 0)   # 4 = VC_GEMM_BF16X3
    assert np.array_equal(flag, entry) and not np.array_equal(f32, flag)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    assert np.abs(f32 - ref).max() < np.abs(flag - ref).max() < TOL * np.abs(ref).max()
    # the ABI-3 shim: a process-wide DEFAULT for calls that carry no flag (never used by the product path)
    assert lib.vc_gemm_get_precision() == 0
    lib.vc_gemm_set_precision(1)
    try:
        assert lib.vc_gemm_get_precision() == 1 and np.array_equal(run(lib.vc_gemm_f32, 0), entry)
    finally:
        lib.vc_gemm_set_precision(0)
    assert np.array_equal(run(lib.vc_gemm_f32, 0), f32)
    from vae_captioning_amd.abi import VaecapError
    with pytest.raises(VaecapError):
        lib.vc_gemm_set_precision(7)


def _small_case(prior="Normal", use_c_v=False):
    from vae_captioning_amd import spec, synth
    from vae_captioning_amd.utils.parameters import Parameters
    p = Parameters()
    p.embed_size, p.encoder_hidden, p.decoder_hidden = 64, 128, 128
    p.latent_size, p.gen_z_samples, p.cnn_feature_size = 20, 6, 256
    p.num_captions, p.batch_size, p.prior, p.use_c_v = 5, 16, prior, use_c_v
    V, B, T = 1000, 16, 12
    rng = np.random.default_rng(0)
    P0 = spec.init_caption_params(p, V, seed=1)
    batch = synth.make_batch(rng, B, p.num_captions, T, V, variable_len=True, feature_size=p.cnn_feature_size, use_ci=spec.uses_ci(p))
    noise = synth.make_noise(rng, B * p.num_captions, T, p)
    return p, V, P0, batch, noise


def test_two_trainers_of_different_precision_coexist_in_one_process(lib):
    """Review item (round 5): precision was process-wide state.  Now an f32 and a bf16x3 Trainer are built side by side and stepped
    ALTERNATELY for three steps; each must be bit-identical to the same trainer run alone in a fresh sequence -- i.e. nothing one does
    reaches the other's arithmetic -- and the two must differ from each other."""
    from vae_captioning_amd.trainer import Trainer
    p, V, P0, batch, noise = _small_case()

    def alone(prec):
        tr = Trainer(p, V, lib=lib, precision=prec)
        tr.load_state_dict(P0)
        tr.set_batch(batch, noise)
        out = []
        for _ in range(3):
            tr.train_step()
            out.append(tr.losses())
        return out, tr.state_dict()
    ref = {prec: alone(prec) for prec in ("f32", "bf16x3")}
    trs = {}
    for prec in ("f32", "bf16x3"):
        trs[prec] = Trainer(p, V, lib=lib, precision=prec)
        trs[prec].load_state_dict(P0)
        trs[prec].set_batch(batch, noise)
    got = {"f32": [], "bf16x3": []}
    for _ in range(3):
        for prec in ("bf16x3", "f32"):
            trs[prec].train_step()
            got[prec].append(trs[prec].losses())
    for prec in ("f32", "bf16x3"):
        assert trs[prec].precision == prec and lib.vc_gemm_get_precision() == 0
        assert got[prec] == ref[prec][0], (prec, got[prec], ref[prec][0])
        sd = trs[prec].state_dict()
        for k, v in ref[prec][1].items():
            assert np.array_equal(sd[k], v), (prec, k)
    assert got["f32"] != got["bf16x3"]


@pytest.mark.parametrize("prior,use_c_v", [("Normal", False), ("AG", True)])
def test_training_step_in_bf16x3_mode_stays_within_the_north_star_tolerance(lib, prior, use_c_v):
    """One whole caption-side step (vae_model/encoder.py:24-110, decoder.py:34-143, main.py:118-177) with every GEMM on the split-bf16
    path, against the fp64 oracle on the same injected tensors: loss / KL within 1e-3 (north_star), gradients within 1e-3 of their
    maximum -- the f32 engine test's bounds."""
    from oracle import caption_model as cm
    from vae_captioning_amd import spec, synth
    from vae_captioning_amd.trainer import Trainer
    from vae_captioning_amd.utils.parameters import Parameters
    p = Parameters()
    p.embed_size, p.encoder_hidden, p.decoder_hidden = 64, 128, 128
    p.latent_size, p.gen_z_samples, p.cnn_feature_size = 20, 6, 256
    p.num_captions, p.batch_size, p.prior, p.use_c_v = 5, 16, prior, use_c_v
    V, B, T = 1000, 16, 12
    rng = np.random.default_rng(0)
    P0 = spec.init_caption_params(p, V, seed=1)
    batch = synth.make_batch(rng, B, p.num_captions, T, V, variable_len=True, feature_size=p.cnn_feature_size, use_ci=spec.uses_ci(p))
    noise = synth.make_noise(rng, B * p.num_captions, T, p)
    f64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
    n64 = f64(noise)
    if prior == "AG":
        from oracle import decode
        n64["c_means"] = decode.init_clusters(90, p.latent_size).astype(np.float64)
    ref = cm.forward_backward(f64(P0), f64(batch), n64, p, global_step=0)
    tr = Trainer(p, V, lib=lib, precision="bf16x3")
    tr.load_state_dict(P0)
    tr.set_batch(batch, noise)
    tr.train_step()
    kld, rec, lb, ann = tr.losses()
    G = tr.cap.grads_dict()
    assert abs(rec - float(ref.rec_loss)) < 1e-3 and abs(kld - float(np.mean(ref.kld))) < 1e-3
    for name, g in ref.grads.items():
        err = np.abs(G[name] - g).max()
        assert err <= 1e-3 * (np.abs(g).max() + 1e-12), (name, err)


@pytest.mark.parametrize("dims", [(3, 320, 64, 512), (2, 1280, 32, 512), (4, 37, 32, 512), (3, 650, 48, 512), (3, 161, 16, 512)],
                         ids=["cfg4-rows", "cfg2-rows-eight-wave-forward", "37-rows", "650-rows", "161-rows"])
def test_lstm_recurrence_kernels_in_bf16x3_mode_match_the_fp64_oracle(lib, dims):
    """The recurrence step kernels (utils/rnn_model.py:23-51 stepped at vae_model/encoder.py:46-55 / decoder.py:100-121) with
    v_mfma_f32_16x16x32_bf16 on split operands -- Wh split by the pack kernels, the row operand split in registers -- and the
    sequence's projection / weight-gradient GEMMs on the split-bf16 GEMM: whole sequences forward and backward against the fp64
    oracle at four times the f32 kernels' tolerances (8e-5 on states and activations, 2e-4 on gradients; the error is the products'
    ~1e-5 carried through T steps).  Rows: four-wave kernel (<= 400), eight-wave forward above, ragged blocks."""
    from .test_gpu_ops import _lstm_seq_check
    _lstm_seq_check(lib, *dims, tol=4.0, kernels=3, bf16x3=True)   # VC_LSTM_KERNELS(3) | VC_LSTM_BF16X3 on the calls themselves


def test_fine_tune_step_in_bf16x3_mode_uses_the_direct_weight_gradient_and_matches_the_oracle(lib, monkeypatch):
    """Trainer(precision="bf16x3") with --fine_tune: the VGG16 weight gradients come from vc_conv3x3_bx_wgrad_f32 (csrc/conv_wgrad_bx.hip)
    -- the step differs bitwise from the one with VC_WGRAD_BX=0 (f32 Winograd weight gradients, everything else equal) -- and every
    gradient stays within 2e-3 (relative l2) of the fp64 oracle evaluated on the device's forward decisions (tests/test_gpu_vgg.py) and
    within 5e-5 of what the f32 Winograd weight gradient gives on the same step."""
    from oracle import caption_model as cm
    from oracle import vgg as ov
    from vae_captioning_amd import spec, synth
    from vae_captioning_amd.trainer import Trainer
    from vae_captioning_amd.utils.parameters import Parameters
    from .test_gpu_vgg import device_cache, rel_l2
    p = Parameters()
    p.fine_tune = True
    p.num_captions, p.gen_z_samples = 2, 4
    V, B, T = 300, 2, 5
    rng = np.random.default_rng(7)
    PC = spec.init_caption_params(p, V, seed=1)
    PV = spec.init_vgg_params(seed=3)
    batch = synth.make_batch(rng, B, p.num_captions, T, V, images=True, variable_len=True)
    noise = synth.make_noise(rng, B * p.num_captions, T, p)
    noise["cnn_drop1"] = (rng.random((B, 4096)) < 0.5).astype(np.float32)
    noise["cnn_drop2"] = (rng.random((B, 4096)) < 0.5).astype(np.float32)
    f64 = lambda d: {k: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in d.items()}
    PV64, PC64, b64, n64 = f64(PV), f64(PC), f64(batch), f64(noise)
    reg = float(ov.l2_reg_loss(PV, p.weight_decay))
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("VC_WGRAD_BX", mode)
        tr = Trainer(p, V, lib=lib, precision="bf16x3")
        tr.load_state_dict({**PC, **PV})
        tr.set_batch(batch, noise)
        tr.train_step()
        torch.cuda.synchronize()
        grads[mode] = {k: v.copy() for k, v in tr.vgg.grads_dict().items()}
        if mode == "1":
            fc2_dev = (tr.vgg.buf["fc2d"] if tr.vgg.keep < 1 else tr.vgg.buf["fc2"]).cpu().numpy().astype(np.float64)
            b64["features"] = fc2_dev
            out = cm.forward_backward(PC64, b64, n64, p, global_step=0, reg_loss=reg)
            GV = ov.backward(PV64, device_cache(tr.vgg, PV64, p.cnn_dropout), out.dfeatures)
    # what bounds the comparison is the gradient that ENTERS the VGG16: the caption side's BPTT in split-bf16 arithmetic carries ~9e-4
    # (cnn/fc2's gradient, which no convolution kernel touches, shows the same figure); the direct weight gradient must add nothing to it
    for n, ref in GV.items():
        e1, e0 = rel_l2(grads["1"][n], ref), rel_l2(grads["0"][n], ref)
        assert e1 < 2e-3 and e1 < e0 + 5e-5, (n, e1, e0)
    changed = [n for n in grads["1"] if "/weights" in n and "/conv" in n and n != "cnn/conv1_1/weights" and not np.array_equal(grads["1"][n], grads["0"][n])]
    assert len(changed) == 12, changed
    for n in changed:   # the two kernels agree to the split-bf16 error
        assert rel_l2(grads["1"][n], grads["0"][n]) < 2e-4, (n, rel_l2(grads["1"][n], grads["0"][n]))
    assert np.array_equal(grads["1"]["cnn/conv1_1/weights"], grads["0"]["cnn/conv1_1/weights"])
