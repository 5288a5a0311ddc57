"""-m gpu: every kernel of libvaecap, called through the C ABI, against the CPU oracle
(oracle/ops.py, oracle/optim.py, oracle/vgg.py) on the same seeded inputs.

Tolerances (fp32 path, stated per test): GEMM-class results are compared with a float64
evaluation of the same contraction and must agree to 2e-6 * sqrt(K) relative to the
tensor max (fp32 accumulation round-off); element-wise kernels to 1e-5 relative (the
device uses __expf / __logf, a few ulp); integer work (Philox words, argmax, counts) is
bit-exact.
"""
import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import optim as OO
from oracle import vgg as OV

from .gpu_util import P, assert_close, dev, empty_bytes, host, stream, zeros

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------------------- GEMM
GEMM_SHAPES = [
    (64, 64, 32), (33, 47, 19), (100, 70, 50), (256, 384, 128), (1280, 2048, 256),
    (300, 256, 4096), (20, 10000, 512), (1280, 256, 15000), (768, 2048, 2560), (1, 4, 4),
    (129, 131, 37),
    (64, 4096, 2048),     # 64-row fc shape: 64 x 128 tiles (plan_gemm skinny), split-K
    (37, 390, 1100),      # skinny with ragged N (390 = 3 x 128 + 6) and K (1100 = 34 x 32 + 12)
    (3200, 512, 5000),    # 100 tiles of 128 x 128: one whole round through split-K (768 / tiles)
    (5000, 640, 2600),    # 200 tiles: below one round, K split into ~1200 workgroups
]


@pytest.mark.parametrize("ta", [0, 1])
@pytest.mark.parametrize("tb", [0, 1])
@pytest.mark.parametrize("shape", GEMM_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_gemm(lib, ta, tb, shape):
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N * 3 + K + ta * 2 + tb)
    A = rng.standard_normal((M, K), dtype=np.float32)
    B = rng.standard_normal((K, N), dtype=np.float32)
    bias = rng.standard_normal(N, dtype=np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64) + bias
    dA = dev(A.T if ta else A)
    dB = dev(B.T if tb else B)
    C = zeros(M, N)
    ws = empty_bytes(lib.vc_gemm_workspace_bytes(M, N, K))
    lib.vc_gemm_f32(stream(), ta, tb, M, N, K, P(dA), M if ta else K, P(dB), K if tb else N, P(C), N, P(dev(bias)), 0,
                    P(ws), ws.numel() * 4)
    assert_close(host(C), ref, 2e-6 * np.sqrt(K) + 1e-6, msg="gemm ta=%d tb=%d %s" % (ta, tb, shape))


def test_gemm_flags_and_ldc(lib):
    M, N, K = 70, 96, 64
    rng = np.random.default_rng(5)
    A = rng.standard_normal((M, K), dtype=np.float32)
    B = rng.standard_normal((K, N), dtype=np.float32)
    C0 = rng.standard_normal((M, N + 8), dtype=np.float32)
    C = dev(C0)
    ws = empty_bytes(lib.vc_gemm_workspace_bytes(M, N, K))
    lib.vc_gemm_f32(stream(), 0, 0, M, N, K, P(dev(A)), K, P(dev(B)), N, P(C), N + 8, None, 3, P(ws), ws.numel() * 4)
    ref = C0.copy()
    ref[:, :N] = np.maximum(C0[:, :N] + A.astype(np.float64) @ B.astype(np.float64), 0)
    assert_close(host(C), ref, 2e-5, msg="gemm relu+accumulate, ldc > N (padding untouched)")


@pytest.mark.parametrize("ta,flags,shape", [(0, 0, (4224, 2052, 96)), (1, 3, (4100, 2052, 100)), (0, 1, (6880, 10000, 512)), (1, 2, (512, 10000, 6880)),
                                            (0, 3, (1000, 2048, 2560)), (1, 0, (256, 2048, 6880))],
                         ids=["unsplit-ragged-N", "KM-ragged-M-K", "logits-fwd", "logits-wgrad-splitK", "splitK-flags", "dWx-splitK"])
def test_gemm_training_shapes(lib, ta, flags, shape):
    """the tb = 0 products of a training step (logits forward / weight gradient, input projections) and ragged M / N / K edges in
    both A storage orders: split-K through the workspace, bias / ReLU / accumulate, ldc > N."""
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
    dA, dB, C = dev(A.T if ta else A), dev(B), dev(C0)
    ws = empty_bytes(lib.vc_gemm_workspace_bytes(M, N, K))
    lib.vc_gemm_f32(stream(), ta, 0, M, N, K, P(dA), M if ta else K, P(dB), N, P(C), N + 4, P(dev(bias)), flags, P(ws), ws.numel() * 4)
    assert_close(host(C), ref, 2e-6 * np.sqrt(K) + 2e-6, msg="gemm ta=%d flags=%d %s (padding column untouched)" % (ta, flags, shape))


@pytest.mark.parametrize("ta,tb,flags", [(0, 0, 0), (0, 1, 1), (1, 0, 2), (1, 1, 3)])
def test_gemm_whole_rounds_tail(lib, ta, tb, flags):
    """776 tiles of 128 x 128 = one round of 768 resident workgroups + one tile row: that row runs as a K-split second
    launch (csrc/gemm.hip plan_gemm, tail_splits) whose reduce applies bias / ReLU / accumulate like the main launch."""
    M, N, K = 12400, 1024, 1024
    assert lib.vc_gemm_workspace_bytes(M, N, K) == 2 * (M - 96 * 128) * N * 4   # 2 splits of 16 K-tiles
    assert lib.vc_gemm_workspace_bytes(M, N, 512) == 0                          # K too short to split: single launch
    rng = np.random.default_rng(9 + flags)
    A = rng.standard_normal((M, K), dtype=np.float32)
    B = rng.standard_normal((K, N), dtype=np.float32)
    bias = rng.standard_normal(N, dtype=np.float32)
    C0 = rng.standard_normal((M, N), dtype=np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64) + bias
    if flags & 2:
        ref = ref + C0
    if flags & 1:
        ref = np.maximum(ref, 0)
    dA, dB, C = dev(A.T if ta else A), dev(B.T if tb else B), dev(C0)
    ws = empty_bytes(lib.vc_gemm_workspace_bytes(M, N, K))
    lib.vc_gemm_f32(stream(), ta, tb, M, N, K, P(dA), M if ta else K, P(dB), K if tb else N, P(C), N, P(dev(bias)), flags, P(ws), ws.numel() * 4)
    assert_close(host(C), ref, 2e-6 * np.sqrt(K) + 2e-6, msg="gemm tail ta=%d tb=%d flags=%d" % (ta, tb, flags))
    C2 = dev(C0)   # no workspace: single launch, same values up to summation order of the tail rows
    lib.vc_gemm_f32(stream(), ta, tb, M, N, K, P(dA), M if ta else K, P(dB), K if tb else N, P(C2), N, P(dev(bias)), flags, None, 0)
    assert_close(host(C2), ref, 2e-6 * np.sqrt(K) + 2e-6, msg="gemm single launch")
    assert torch.equal(C[:96 * 128], C2[:96 * 128]) and not torch.equal(C[96 * 128:], C2[96 * 128:])


def test_gemm_rejects_bad_arguments(lib):
    from vae_captioning_amd.abi import VaecapError
    with pytest.raises(VaecapError):
        lib.vc_gemm_f32(stream(), 0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, None, 0)
    x = zeros(1280, 15000)
    w = zeros(15000, 256)
    c = zeros(1280, 256)
    with pytest.raises(VaecapError):  # split-K without workspace
        lib.vc_gemm_f32(stream(), 0, 0, 1280, 256, 15000, P(x), 15000, P(w), 256, P(c), 256, None, 0, None, 0)


# ----------------------------------------------------------------------------- LSTM
def _lstm_case(T, N, E, H, seed, full_len=False):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((T, N, E), dtype=np.float32)
    W = (rng.standard_normal((E + H, 4 * H), dtype=np.float32) * np.float32(1.0 / np.sqrt(E + H)))
    b = rng.standard_normal(4 * H, dtype=np.float32) * np.float32(0.1)
    lens = np.full(N, T, np.int32) if full_len else rng.integers(0, T + 1, size=N).astype(np.int32)
    if not full_len:
        lens[0] = T
        lens[1] = 0
    return X, W, b, lens


@pytest.mark.parametrize("dims", [(6, 70, 32, 64, False), (4, 700, 48, 96, False), (3, 1280, 256, 512, True), (5, 330, 64, 512, True),
                                  (4, 650, 32, 1024, True)],
                         ids=["small", "mid-128rows", "cfg2-shape", "cfg4-rows-ragged", "H1024"])
def test_lstm_seq_fwd_bwd(lib, dims):
    T, N, E, H, full = dims
    _lstm_seq_check(lib, T, N, E, H)   # flags 0 = auto: the recurrence kernels at H = 512 (forward up to 640 rows), the split form elsewhere


@pytest.mark.parametrize("dims", [(3, 320, 64, 512), (2, 1280, 32, 512), (4, 37, 32, 512), (3, 650, 48, 512), (2, 5, 16, 512), (3, 161, 16, 512)],
                         ids=["cfg4-rows", "cfg2-rows-4-passes", "37-rows", "650-rows", "5-rows", "161-rows"])
def test_lstm_recurrence_kernels(lib, dims):
    """mode 3: the register-operand recurrence kernels (16x16x4 MFMA, K split over four waves) forward AND backward, on whole and
    ragged row blocks, one and several passes per workgroup"""
    _lstm_seq_check(lib, *dims, kernels=3)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_lstm_sequence_modes_agree_with_oracle(lib, mode):
    """every other driver mode (round-1 fused kernels, GEMM + gate kernels, auto) on the cfg4 row count at H = 512"""
    _lstm_seq_check(lib, 3, 320, 64, 512, kernels=mode)


def test_lstm_deprecated_process_wide_mode_still_selects_for_calls_without_a_choice(lib):
    """vc_lstm_set_mode (ABI <= 3) remains the default of calls whose flags choose nothing; a call's own VC_LSTM_KERNELS wins over it"""
    lib.vc_lstm_set_mode(1)
    try:
        _lstm_seq_check(lib, 2, 320, 32, 512)              # runs the GEMM + gate form
        _lstm_seq_check(lib, 2, 320, 32, 512, kernels=3)   # the call's own choice
    finally:
        lib.vc_lstm_set_mode(2)


def _lstm_seq_check(lib, T, N, E, H, tol=1.0, kernels=None, bf16x3=False):
    fl = (0 if kernels is None else kernels + 1) | (0x10 if bf16x3 else 0)   # VC_LSTM_KERNELS(k) | VC_LSTM_BF16X3
    X, W, b, lens = _lstm_case(T, N, E, H, seed=T * N)
    rng = np.random.default_rng(1)
    dhs = rng.standard_normal((T + 1, N, H), dtype=np.float32) * np.float32(0.1)
    dhs[0] = 0
    cache = O.lstm_seq_fwd(X.astype(np.float64), lens, W.astype(np.float64), b.astype(np.float64))
    rdX, rdW, rdb, _, _ = O.lstm_seq_bwd(cache, dhs.astype(np.float64))

    dX_, dW_, db_ = zeros(T, N, E), zeros(E + H, 4 * H), zeros(4 * H)
    act, cs, hs = zeros(T, N, 4 * H), zeros(T + 1, N, H), zeros(T + 1, N, H)
    ws = empty_bytes(lib.vc_lstm_seq_workspace_bytes(T, N, E, H))
    tX, tW, tb, tl = dev(X), dev(W), dev(b), dev(lens)
    lib.vc_lstm_seq_fwd_f32(stream(), T, N, E, H, P(tX), P(tW), P(tb), P(tl), P(act), P(cs), P(hs), P(ws), ws.numel() * 4, fl)
    assert_close(host(hs), cache["hs"], 2e-5 * tol, msg="lstm hs")
    assert_close(host(cs), cache["cs"], 2e-5 * tol, msg="lstm cs")
    assert_close(host(act), cache["act"], 2e-5 * tol, msg="lstm gate activations")
    dH, dC, dG = dev(dhs[T]).clone(), zeros(N, H), zeros(T, N, 4 * H)
    text = dev(dhs)
    # dH_run starts as the gradient w.r.t. the final state; dhs_ext[T] must then not be double counted
    text[T].zero_()
    lib.vc_lstm_seq_bwd_f32(stream(), T, N, E, H, P(tX), P(tW), P(tl), P(act), P(cs), P(hs), P(text), P(dH), P(dC), P(dG),
                            P(dX_), P(dW_), P(db_), P(ws), ws.numel() * 4, fl)
    assert_close(host(dX_), rdX, 5e-5 * tol, msg="lstm dX")
    assert_close(host(dW_), rdW, 5e-5 * tol, msg="lstm dW")
    assert_close(host(db_), rdb, 5e-5 * tol, msg="lstm db")


# ----------------------------------------------------------------------------- embedding
def test_embedding_gather_scatter(lib):
    rng = np.random.default_rng(3)
    V, E, R = 1000, 256, 5000
    table = rng.standard_normal((V, E), dtype=np.float32)
    ids = rng.integers(0, V, size=R).astype(np.int32)
    ids[:50] = 7
    out = zeros(R, E)
    lib.vc_embedding_gather_f32(stream(), P(dev(table)), P(dev(ids)), R, E, V, P(out))
    np.testing.assert_array_equal(host(out), table[ids])  # bit-exact copy
    dX = rng.standard_normal((R, E), dtype=np.float32)
    dt = zeros(V, E)
    lib.vc_embedding_scatter_add_f32(stream(), P(dt), P(dev(ids)), R, E, V, P(dev(dX)))
    assert_close(host(dt), O.embedding_bwd(V, ids, dX.astype(np.float64)), 1e-5, msg="scatter_add")
    order = np.argsort(ids, kind="stable").astype(np.int32)
    seg = np.concatenate([[0], np.cumsum(np.bincount(ids, minlength=V))]).astype(np.int32)
    runs = []
    for _ in range(2):
        d2 = torch.full((V, E), 7.0, device="cuda")  # every row must be overwritten
        lib.vc_embedding_grad_sorted_f32(stream(), P(d2), P(dev(order)), P(dev(seg)), E, V, P(dev(dX)))
        runs.append(host(d2))
    assert_close(runs[0], O.embedding_bwd(V, ids, dX.astype(np.float64)), 1e-6, msg="sorted embedding grad")
    np.testing.assert_array_equal(runs[0], runs[1])  # deterministic
    tch = zeros(V)
    lib.vc_mark_rows_f32(stream(), P(tch), P(dev(ids)), R, V)
    ref = np.zeros(V, np.float32)
    ref[ids] = 1
    np.testing.assert_array_equal(host(tch), ref)
    # unaligned width -> scalar path
    E2 = 150
    t2 = rng.standard_normal((V, E2), dtype=np.float32)
    o2 = zeros(R, E2)
    lib.vc_embedding_gather_f32(stream(), P(dev(t2)), P(dev(ids)), R, E2, V, P(o2))
    np.testing.assert_array_equal(host(o2), t2[ids])


# ----------------------------------------------------------------------------- softmax-CE
@pytest.mark.parametrize("V", [10000, 1003, 11313, 64])
def test_softmax_xent(lib, V):
    rng = np.random.default_rng(V)
    R = 300
    logits = (rng.standard_normal((R, V), dtype=np.float32) * 3).astype(np.float32)
    labels = rng.integers(1, V, size=R).astype(np.int32)
    labels[::7] = 0  # PAD rows (quirk Q8)
    loss, cache = O.xent_masked_fwd(logits.astype(np.float64), labels)
    gscale = 3.0
    dref = O.xent_masked_bwd(cache, gscale)
    tl = dev(logits)
    den, rl = zeros(1), zeros(R)
    lib.vc_count_nonzero_i32(stream(), P(dev(labels)), R, P(den))
    assert host(den)[0] == float((labels != 0).sum())
    # forward only must leave the logits untouched
    lib.vc_softmax_xent_f32(stream(), P(tl), P(dev(labels)), R, V, V, P(den), gscale, P(rl), 0)
    np.testing.assert_array_equal(host(tl), logits)
    lib.vc_softmax_xent_f32(stream(), P(tl), P(dev(labels)), R, V, V, P(den), gscale, P(rl), 1)
    tot = zeros(1)
    lib.vc_reduce_sum_f32(stream(), P(rl), R, 1.0, P(tot), 0)
    np.testing.assert_allclose(host(tot)[0] / host(den)[0], loss, rtol=1e-5)
    assert_close(host(tl), dref, 1e-5, msg="dlogits V=%d" % V)
    assert np.all(host(tl)[labels == 0] == 0)
    probs = zeros(R, V)
    lib.vc_softmax_rows_f32(stream(), P(dev(logits)), R, V, V, P(probs), V)
    assert_close(host(probs), cache["p"], 1e-5, msg="softmax rows")
    am = torch.zeros(R, dtype=torch.int32, device="cuda")
    lib.vc_argmax_rows_f32(stream(), P(dev(logits)), R, V, V, P(am))
    np.testing.assert_array_equal(host(am), logits.argmax(axis=1))


def test_argmax_tie_breaks_to_lowest_index(lib):
    x = np.zeros((3, 700), np.float32)
    x[0, [5, 300, 699]] = 2.0
    x[1, :] = -1.0
    x[2, 698] = 1.0
    am = torch.zeros(3, dtype=torch.int32, device="cuda")
    lib.vc_argmax_rows_f32(stream(), P(dev(x)), 3, 700, 700, P(am))
    np.testing.assert_array_equal(host(am), [5, 0, 698])


# ----------------------------------------------------------------------------- latent
@pytest.mark.parametrize("mode", [0, 1])
def test_latent_sample_kl_bwd(lib, mode):
    rng = np.random.default_rng(11 + mode)
    S, N, L = 7, 50, 150
    mean = rng.standard_normal((N, L), dtype=np.float32) * np.float32(0.3)
    std = np.exp(rng.standard_normal((N, L), dtype=np.float32) * np.float32(0.3)).astype(np.float32)
    eps = rng.standard_normal((S, N, L), dtype=np.float32)
    dz = rng.standard_normal((S, N, L), dtype=np.float32)
    mu_p = rng.standard_normal((N, L), dtype=np.float32) * np.float32(0.1)
    z = zeros(S, N, L)
    lib.vc_latent_sample_f32(stream(), S, N, L, P(dev(mean)), P(dev(std)), P(dev(eps)), P(z))
    assert_close(host(z), O.sample_z_fwd(mean, std, eps), 1e-6, msg="z")
    rk = zeros(N)
    lib.vc_kl_rows_f32(stream(), N, L, mode, P(dev(mean)), P(dev(std)), P(dev(mu_p)), P(rk))
    m64, s64 = mean.astype(np.float64), std.astype(np.float64)
    ann = 0.37
    if mode == 0:
        ref_rows = -0.5 * (1 + np.log(s64 ** 2 + 1e-5) - m64 ** 2 - s64 ** 2).sum(1)
        km, ks = O.kl_normal_bwd(m64, s64, ann / 10)
        kl_scale = 0.1 / N
    else:
        # kl_ag_* take (c_i, cluster_means); c_i = I, means = mu_p gives c_i @ means == mu_p
        ref_rows = O.kl_ag_fwd(m64, s64, np.eye(N), mu_p.astype(np.float64))
        km, ks = O.kl_ag_bwd(m64, s64, np.eye(N), mu_p.astype(np.float64), np.full(N, ann / 10))
        kl_scale = 0.1
    assert_close(host(rk), ref_rows, 1e-5, msg="kl rows mode %d" % mode)
    dm_ref, ds_ref = O.sample_z_bwd(dz.astype(np.float64), eps.astype(np.float64))
    dm_ref, ds_ref = dm_ref + km, ds_ref + ks
    dm, ds = zeros(N, L), zeros(N, L)
    annd = dev(np.array([ann], np.float32))
    lib.vc_latent_bwd_f32(stream(), S, N, L, mode, 0, P(dev(dz)), P(dev(eps)), P(dev(mean)), P(dev(std)), P(dev(mu_p)),
                          P(annd), kl_scale, P(dm), P(ds))
    assert_close(host(dm), dm_ref, 1e-5, msg="dmean")
    assert_close(host(ds), ds_ref, 1e-5, msg="dstd")
    lib.vc_latent_bwd_f32(stream(), S, N, L, mode, 1, P(dev(dz)), P(dev(eps)), P(dev(mean)), P(dev(std)), P(dev(mu_p)),
                          P(annd), kl_scale, P(dm), P(ds))
    assert_close(host(ds), ds_ref * s64, 1e-5, msg="dlogstd")


@pytest.mark.parametrize("gmm", [False, True])
def test_heads_mix(lib, gmm):
    rng = np.random.default_rng(21)
    N, K, L = 37, 90, 150
    heads = rng.standard_normal((N, 2 * K * L), dtype=np.float32) * np.float32(0.2)
    ci = np.zeros((N, K), np.float32)
    for n in range(N):
        ci[n, rng.choice(K, size=3, replace=False)] = 1 / 3
    idx = rng.integers(0, K, size=N).astype(np.int32)
    tm = heads[:, :K * L].reshape(N, K, L).astype(np.float64)
    etl = np.exp(heads[:, K * L:].reshape(N, K, L).astype(np.float64))
    if gmm:
        mref, sref = tm[np.arange(N), idx], etl[np.arange(N), idx]
    else:
        mref, sref = np.einsum("nk,nkl->nl", ci, tm), np.einsum("nk,nkl->nl", ci, etl)
    mean, std = zeros(N, L), zeros(N, L)
    lib.vc_heads_mix_fwd_f32(stream(), N, K, L, P(dev(heads)), None if gmm else P(dev(ci)), P(dev(idx)) if gmm else None,
                             P(mean), P(std))
    assert_close(host(mean), mref, 1e-5, msg="mix mean")
    assert_close(host(std), sref, 1e-5, msg="mix std")
    dmean = rng.standard_normal((N, L), dtype=np.float32)
    dstd = rng.standard_normal((N, L), dtype=np.float32)
    dh = zeros(N, 2 * K * L)
    lib.vc_heads_mix_bwd_f32(stream(), N, K, L, P(dev(heads)), None if gmm else P(dev(ci)), P(dev(idx)) if gmm else None,
                             P(dev(dmean)), P(dev(dstd)), P(dh))
    w = np.eye(K)[idx] if gmm else ci.astype(np.float64)
    ref = np.concatenate([(w[:, :, None] * dmean[:, None, :]).reshape(N, -1),
                          (w[:, :, None] * dstd[:, None, :] * etl).reshape(N, -1)], axis=1)
    assert_close(host(dh), ref, 1e-5, msg="mix bwd")


# ----------------------------------------------------------------------------- small ops
def test_small_ops(lib):
    rng = np.random.default_rng(31)
    R, C = 5000, 300
    x = rng.standard_normal((R, C), dtype=np.float32)
    out = zeros(C)
    ws = empty_bytes(lib.vc_colsum_workspace_bytes(R, C))
    lib.vc_colsum_f32(stream(), P(dev(x)), R, C, C, P(out), 0, P(ws), ws.numel() * 4)
    assert_close(host(out), x.astype(np.float64).sum(0), 1e-5, msg="colsum")
    lib.vc_colsum_f32(stream(), P(dev(x)), R, C, C, P(out), 1, P(ws), ws.numel() * 4)
    assert_close(host(out), 2 * x.astype(np.float64).sum(0), 1e-5, msg="colsum accumulate")
    mask = (rng.random((R, C)) < 0.7).astype(np.float32)
    y = zeros(R, C)
    lib.vc_dropout_f32(stream(), P(dev(x)), P(dev(mask)), 0.7, R * C, P(y))
    assert_close(host(y), O.dropout_fwd(x, mask, 0.7), 1e-6, msg="dropout")
    act = np.maximum(rng.standard_normal((R, C), dtype=np.float32), 0)
    lib.vc_relu_bwd_f32(stream(), P(dev(x)), P(dev(act)), P(dev(mask)), 0.5, R * C, P(y))
    assert_close(host(y), x * (act > 0) * mask / 0.5, 1e-6, msg="relu_bwd + dropout")
    lib.vc_relu_bwd_f32(stream(), P(dev(x)), P(dev(act)), None, 1.0, R * C, P(y))
    np.testing.assert_array_equal(host(y), x * (act > 0))
    B, nc, E = 33, 5, 256
    f = rng.standard_normal((B, E), dtype=np.float32)
    t = zeros(B * nc, E)
    lib.vc_tile_rows_f32(stream(), P(dev(f)), B, nc, E, P(t))
    np.testing.assert_array_equal(host(t), np.repeat(f, nc, axis=0))
    g = rng.standard_normal((B * nc, E), dtype=np.float32)
    s = zeros(B, E)
    lib.vc_segment_sum_rows_f32(stream(), P(dev(g)), B, nc, E, P(s), 0)
    assert_close(host(s), g.reshape(B, nc, E).sum(1), 1e-6, msg="segment sum")
    e = zeros(R, C)
    lib.vc_exp_f32(stream(), P(dev(x)), R * C, P(e))
    assert_close(host(e), np.exp(x.astype(np.float64)), 1e-5, msg="exp")


# ----------------------------------------------------------------------------- optimisers
def test_clip_and_adam_sgd_momentum(lib):
    rng = np.random.default_rng(41)
    n = 100003
    p0 = rng.standard_normal(n, dtype=np.float32)
    nb = lib.vc_sumsq_blocks()
    part = zeros(2 * nb)
    g1 = rng.standard_normal(n, dtype=np.float32) * np.float32(0.05)
    extra = rng.standard_normal(5000, dtype=np.float32)
    tg = dev(g1)
    lib.vc_sumsq_partial_f32(stream(), P(tg), n, P(part))
    lib.vc_sumsq_partial_f32(stream(), P(dev(extra)), extra.size, part.data_ptr() + nb * 4)
    ns = zeros(2)
    lib.vc_clip_finalize_f32(stream(), P(part), 2 * nb, 5.0, P(ns))
    norm = np.sqrt((g1.astype(np.float64) ** 2).sum() + (extra.astype(np.float64) ** 2).sum())
    np.testing.assert_allclose(host(ns)[0], norm, rtol=1e-5)
    np.testing.assert_allclose(host(ns)[1], OO.clip_scale(norm, 5.0), rtol=1e-5)
    assert host(ns)[1] < 1.0
    # Adam, 3 steps, device-resident step scalars
    Pn = {"w": p0.copy()}
    st = {}
    tp, tm, tv = dev(p0), zeros(n), zeros(n)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    sc = zeros(8)
    for t in range(1, 4):
        g = rng.standard_normal(n, dtype=np.float32) * np.float32(0.05)
        lib.vc_step_update(stream(), P(step), P(sc), 5e-4, 1e-5, 0.8, 0.999, 2.0, 1, 100)
        lib.vc_adam_f32(stream(), P(tp), P(dev(g)), P(tm), P(tv), n, P(sc), sc.data_ptr() + 5 * 4, 0.8, 0.999, 1e-8, 0.0)
        OO.adam_step(Pn, {"w": g}, st, 5e-4, t)
    # scalars[5] is zero -> scale 0 would kill the gradient: it must have been read as the clip scale
    assert np.all(host(tp) == p0), "scale pointer ignored"
    tp, tm, tv = dev(p0), zeros(n), zeros(n)
    step.zero_()
    one = dev(np.array([0.5], np.float32))
    rng = np.random.default_rng(42)
    Pn = {"w": p0.copy()}
    st = {}
    for t in range(1, 4):
        g = rng.standard_normal(n, dtype=np.float32) * np.float32(0.05)
        lib.vc_step_update(stream(), P(step), P(sc), 5e-4, 1e-5, 0.8, 0.999, 2.0, 1, 100)
        lib.vc_adam_f32(stream(), P(tp), P(dev(g)), P(tm), P(tv), n, P(sc), P(one), 0.8, 0.999, 1e-8, 4e-5)
        OO.adam_step(Pn, {"w": g}, st, 5e-4, t, scale=0.5, l2=4e-5)
    assert_close(host(tp), Pn["w"], 2e-6, msg="adam params after 3 steps")
    assert host(step)[0] == 3
    s = host(sc)
    np.testing.assert_allclose(s[1], (np.tanh((2 - 2000.0) / 1000) + 1) / 2, rtol=1e-4)
    np.testing.assert_allclose(s[2], 5e-4, rtol=1e-6)
    # SGD + Momentum (row-masked)
    V, E = 50, 16
    w0 = rng.standard_normal((V, E), dtype=np.float32)
    g = rng.standard_normal((V, E), dtype=np.float32)
    lr = dev(np.array([0.1], np.float32))
    tw = dev(w0)
    lib.vc_sgd_f32(stream(), P(tw), P(dev(g)), V * E, P(lr), None, 0.0)
    assert_close(host(tw), w0 - np.float32(0.1) * g, 1e-6, msg="sgd")
    touched = np.zeros(V, np.float32)
    touched[[1, 4, 9]] = 1
    tw, ta = dev(w0), zeros(V, E)
    Pn, st = {"e": w0.copy()}, {}
    for _ in range(2):
        lib.vc_momentum_f32(stream(), P(tw), P(dev(g)), P(ta), V * E, P(lr), None, 0.9, 0.0, P(dev(touched)), E)
        OO.momentum_step(Pn, {"e": g}, st, 0.1, touched={"e": touched.astype(bool)})
    assert_close(host(tw), Pn["e"], 1e-6, msg="momentum (sparse rows)")


def _philox_ref(n, seed, offset):
    """Philox4x32-10 (Salmon et al.), counter = (i/4 lo, i/4 hi, offset lo, offset hi)."""
    M0, M1 = 0xD2511F53, 0xCD9E8D57
    out = np.zeros(((n + 3) // 4) * 4, np.uint32)
    for q in range((n + 3) // 4):
        c = [q & 0xffffffff, q >> 32, offset & 0xffffffff, offset >> 32]
        k = [seed & 0xffffffff, seed >> 32]
        for _ in range(10):
            p0, p1 = M0 * c[0], M1 * c[2]
            c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xffffffff, p1 & 0xffffffff, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xffffffff,
                 p0 & 0xffffffff]
            k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
        out[4 * q:4 * q + 4] = c
    return out[:n]


def test_philox_bit_exact_and_moments(lib):
    n = 1001
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    lib.vc_philox_u32(stream(), P(out), n, 0x123456789abcdef, 77, None)
    np.testing.assert_array_equal(host(out).view(np.uint32), _philox_ref(n, 0x123456789abcdef, 77))
    # known-answer from the Random123 test vectors: counter 0, key 0
    z = torch.zeros(4, dtype=torch.int32, device="cuda")
    lib.vc_philox_u32(stream(), P(z), 4, 0, 0, None)
    np.testing.assert_array_equal(host(z).view(np.uint32), np.array([0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8], np.uint32))
    m = 4_000_000
    x = zeros(m)
    lib.vc_philox_normal_f32(stream(), P(x), m, 9, 0, None)
    h = host(x).astype(np.float64)
    assert abs(h.mean()) < 3e-3 and abs(h.std() - 1) < 3e-3 and abs((h ** 4).mean() - 3) < 0.05
    y = zeros(m)
    lib.vc_philox_normal_f32(stream(), P(y), m, 9, 0, None)
    np.testing.assert_array_equal(host(x), host(y))  # counter-based: reproducible
    st = torch.tensor([5], dtype=torch.int32, device="cuda")
    lib.vc_philox_normal_f32(stream(), P(y), m, 9, 0, P(st))
    assert not np.array_equal(host(x), host(y))
    lib.vc_philox_bernoulli_f32(stream(), P(y), m, 0.7, 3, 0, None)
    assert abs(host(y).mean() - 0.7) < 2e-3 and set(np.unique(host(y))) == {0.0, 1.0}


def test_philox_streams_do_not_alias_across_steps(lib):
    """The engine keeps a stream id in offset >> 32 (eps 1, drop_in 2, drop_out 3, fc masks 8 / 9) and advances a device
    step counter.  step is part of the KEY: stream k at step s+1 must not repeat stream k+1 at step s (it did while
    step shared the high counter word with the stream id)."""
    n, seed = 4096, 1234 * 1000003
    draws = {}
    for sid in (1, 2, 3, 8, 9):
        for s in (0, 1, 2):
            out = torch.zeros(n, dtype=torch.int32, device="cuda")
            st = torch.tensor([s], dtype=torch.int32, device="cuda")
            lib.vc_philox_u32(stream(), P(out), n, seed, sid << 32, P(st))
            draws[(sid, s)] = host(out).view(np.uint32).copy()
    keys = sorted(draws)
    for i, a in enumerate(keys):
        for b in keys[i + 1:]:
            assert (draws[a] == draws[b]).mean() < 0.01, (a, b)
    # bit-exact definition: key = (seed lo, seed hi + step), counter = (quad, offset)
    np.testing.assert_array_equal(draws[(2, 1)][:64], _philox_ref(64, seed + (1 << 32), 2 << 32))


# ----------------------------------------------------------------------------- VGG kernels
CONV_CASES = [(2, 8, 6, 4, 8), (1, 14, 14, 64, 128), (2, 12, 10, 64, 64), (1, 7, 7, 512, 512), (3, 16, 16, 4, 64),
              (1, 28, 28, 128, 256), (1, 10, 10, 64, 256)]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv3x3_fwd_dgrad_wgrad(lib, case):
    B, H, W, Ci, Co = case
    rng = np.random.default_rng(sum(case))
    x = np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)
    w = rng.standard_normal((3, 3, Ci, Co), dtype=np.float32) * np.float32(1 / np.sqrt(9 * Ci))
    b = rng.standard_normal(Co, dtype=np.float32)
    dy = rng.standard_normal((B, H, W, Co), dtype=np.float32)
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    yref = np.maximum(OV.conv3x3_fwd(x64, w64, b.astype(np.float64)), 0)
    dxref, dwref, dbref = OV.conv3x3_bwd(x64, w64, dy.astype(np.float64))
    tx, tw, tdy = dev(x), dev(w), dev(dy)
    y = zeros(B, H, W, Co)
    lib.vc_conv3x3_fwd_f32(stream(), B, H, W, Ci, Co, P(tx), P(tw), P(dev(b)), P(y), 1, None, 0)
    assert_close(host(y), yref, 2e-6 * np.sqrt(9 * Ci) + 1e-6, msg="conv fwd")
    dx = zeros(B, H, W, Ci)
    lib.vc_conv3x3_dgrad_f32(stream(), B, H, W, Ci, Co, P(tdy), P(tw), P(tx), P(dx), None, 0)
    assert_close(host(dx), dxref * (x > 0), 2e-6 * np.sqrt(9 * Co) + 1e-6, msg="conv dgrad (+relu mask)")
    lib.vc_conv3x3_dgrad_f32(stream(), B, H, W, Ci, Co, P(tdy), P(tw), None, P(dx), None, 0)
    assert_close(host(dx), dxref, 2e-6 * np.sqrt(9 * Co) + 1e-6, msg="conv dgrad")
    dw = zeros(3, 3, Ci, Co)
    ws = empty_bytes(lib.vc_conv3x3_wgrad_workspace_bytes(B, H, W, Ci, Co))
    db = zeros(Co)
    lib.vc_conv3x3_wgrad_f32(stream(), B, H, W, Ci, Co, P(tx), P(tdy), P(dw), P(db), 0, P(ws), ws.numel() * 4)
    assert_close(host(dw), dwref, 2e-6 * np.sqrt(B * H * W) + 1e-6, msg="conv wgrad")
    assert_close(host(db), dbref, 1e-5, msg="conv bias grad (fused into wgrad)")
    lib.vc_conv3x3_wgrad_f32(stream(), B, H, W, Ci, Co, P(tx), P(tdy), P(dw), None, 1, P(ws), ws.numel() * 4)
    assert_close(host(dw), 2 * dwref, 2e-6 * np.sqrt(B * H * W) + 1e-6, msg="conv wgrad accumulate, no bias")
    db2 = zeros(Co)
    ws2 = empty_bytes(lib.vc_colsum_workspace_bytes(B * H * W, Co))
    lib.vc_colsum_f32(stream(), P(tdy), B * H * W, Co, Co, P(db2), 0, P(ws2), ws2.numel() * 4)
    assert_close(host(db2), dbref, 1e-5, msg="colsum")


@pytest.mark.parametrize("case", [(2, 224, 224, 64, 128, "fwd"), (2, 224, 224, 128, 64, "dgrad"), (8, 224, 224, 64, 64, "fwd"),
                                  (1, 224, 320, 128, 256, "fwd"), (3, 14, 14, 512, 512, "both")], ids=lambda c: "x".join(map(str, c)))
def test_conv_whole_rounds_tail_split(lib, case):
    """Forward / data-gradient launches whose tile count is a whole number of rounds plus a few tiles run the
    remainder as a K-split second launch (csrc/conv.hip launch_rounds): same values as the single launch up to the
    summation order of the split rows, and the forward still matches the fp64 oracle on a sample of rows."""
    B, H, W, Ci, Co, which = case
    rng = np.random.default_rng(B + Ci)
    x = torch.from_numpy(np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)).cuda()
    w = torch.from_numpy(rng.standard_normal((3, 3, Ci, Co), dtype=np.float32) * np.float32(1 / np.sqrt(9 * Ci))).cuda()
    b = torch.from_numpy(rng.standard_normal(Co, dtype=np.float32)).cuda()
    dy = torch.from_numpy(rng.standard_normal((B, H, W, Co), dtype=np.float32)).cuda()
    if which in ("fwd", "both"):
        nb = lib.vc_conv3x3_fwd_workspace_bytes(B, H, W, Ci, Co)
        assert nb > 0, "this shape must trigger the tail split"
        ws = empty_bytes(nb)
        y0, y1 = zeros(B, H, W, Co), zeros(B, H, W, Co)
        lib.vc_conv3x3_fwd_f32(stream(), B, H, W, Ci, Co, P(x), P(w), P(b), P(y0), 1, None, 0)
        lib.vc_conv3x3_fwd_f32(stream(), B, H, W, Ci, Co, P(x), P(w), P(b), P(y1), 1, P(ws), ws.numel() * 4)
        assert_close(host(y1), host(y0), 4e-6 * np.sqrt(9 * Ci) + 1e-6, msg="fwd tail split vs single launch")
        assert not torch.equal(y0, y1) or Ci * 9 <= 128      # the tail rows really took the split path
        x64 = x[-1:, -6:].cpu().numpy().astype(np.float64)    # last rows of the last image: inside the tail
        yref = np.maximum(OV.conv3x3_fwd(x64, w.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)), 0)
        assert_close(host(y1)[-1, -5:], yref[0, 1:], 2e-6 * np.sqrt(9 * Ci) + 1e-6, msg="fwd tail rows vs oracle")
    if which in ("dgrad", "both"):
        nb = lib.vc_conv3x3_dgrad_workspace_bytes(B, H, W, Ci, Co)
        assert nb > 0, "this shape must trigger the tail split"
        ws = empty_bytes(nb)
        d0, d1 = zeros(B, H, W, Ci), zeros(B, H, W, Ci)
        for mask in (P(x), None):
            lib.vc_conv3x3_dgrad_f32(stream(), B, H, W, Ci, Co, P(dy), P(w), mask, P(d0), None, 0)
            lib.vc_conv3x3_dgrad_f32(stream(), B, H, W, Ci, Co, P(dy), P(w), mask, P(d1), P(ws), ws.numel() * 4)
            assert_close(host(d1), host(d0), 4e-6 * np.sqrt(9 * Co) + 1e-6, msg="dgrad tail split vs single launch")


def test_conv_workspace_too_small_means_single_launch_and_beam_argument_checks(lib):
    from vae_captioning_amd.abi import VaecapError
    B, H, W, Ci, Co = 2, 224, 224, 64, 128
    need = lib.vc_conv3x3_fwd_workspace_bytes(B, H, W, Ci, Co)
    assert need > 0
    x, w, b = zeros(B, H, W, Ci) + 1.0, zeros(3, 3, Ci, Co) + 0.01, zeros(Co)
    y0, y1 = zeros(B, H, W, Co), zeros(B, H, W, Co)
    ws = empty_bytes(need)
    lib.vc_conv3x3_fwd_f32(stream(), B, H, W, Ci, Co, P(x), P(w), P(b), P(y0), 1, None, 0)
    lib.vc_conv3x3_fwd_f32(stream(), B, H, W, Ci, Co, P(x), P(w), P(b), P(y1), 1, P(ws), need - 4)   # one float short: not used
    assert torch.equal(y0, y1)
    i32 = dict(dtype=torch.int32, device="cuda")
    d = torch.zeros(64, dtype=torch.float64, device="cuda")
    z = torch.zeros(4096, **i32)
    f = zeros(64)
    with pytest.raises(VaecapError):   # beam sizes above 16 are rejected, not truncated
        lib.vc_beam_update(stream(), 1, 17, 8, 2, 0.7, P(f), P(z), P(z), P(z), P(d), P(d), P(z), P(z), P(z), P(d), P(d), P(z), P(z), P(z), P(z), P(z), P(z))
    with pytest.raises(VaecapError):
        lib.vc_beam_update(stream(), 1, 2, 8, 2, 0.7, None, P(z), P(z), P(z), P(d), P(d), P(z), P(z), P(z), P(d), P(d), P(z), P(z), P(z), P(z), P(z), P(z))


def test_maxpool_and_preprocess(lib):
    rng = np.random.default_rng(51)
    B, H, W, C = 2, 8, 12, 64
    x = np.maximum(rng.standard_normal((B, H, W, C), dtype=np.float32), 0)
    x[0, 0:2, 0:2, :8] = 0.0       # an all-zero window (tie at zero)
    x[1, 2, 2, :] = x[1, 2, 3, :] = 5.0  # a positive tie: first in scan order wins
    yref, arg = OV.maxpool_fwd(x)
    y = zeros(B, H // 2, W // 2, C)
    lib.vc_maxpool2x2_fwd_f32(stream(), B, H, W, C, P(dev(x)), P(y))
    np.testing.assert_array_equal(host(y), yref)
    dy = rng.standard_normal(yref.shape).astype(np.float32)
    dx = zeros(B, H, W, C)
    lib.vc_maxpool2x2_bwd_f32(stream(), B, H, W, C, P(dev(x)), P(dev(dy)), P(dx), 0)
    np.testing.assert_array_equal(host(dx), OV.maxpool_bwd(dy, arg, x.shape))
    lib.vc_maxpool2x2_bwd_f32(stream(), B, H, W, C, P(dev(x)), P(dev(dy)), P(dx), 1)
    np.testing.assert_array_equal(host(dx), OV.maxpool_bwd(dy, arg, x.shape) * (x > 0))
    img = rng.integers(0, 256, size=(2, 10, 6, 3)).astype(np.float32)
    o = zeros(2, 10, 6, 4)
    lib.vc_vgg_preprocess_f32(stream(), P(dev(img)), 2, 10, 6, P(o))
    ref = np.concatenate([img - OV.MEAN_RGB, np.zeros((2, 10, 6, 1), np.float32)], axis=3)
    np.testing.assert_array_equal(host(o), ref)
    # the uint8 entry (the reference's HDF5 pixels, preprocess.py:27-28): bit-identical to the float32 feed of the same pixels
    o8 = zeros(2, 10, 6, 4)
    lib.vc_vgg_preprocess_u8(stream(), P(dev(img.astype(np.uint8))), 2, 10, 6, P(o8))
    np.testing.assert_array_equal(host(o8), ref)
    w3 = rng.standard_normal((3, 3, 3, 64), dtype=np.float32)
    w4 = zeros(3, 3, 4, 64)
    lib.vc_pad_dim_f32(stream(), P(dev(w3)), 9, 3, 4, 64, P(w4))
    ref4 = np.zeros((3, 3, 4, 64), np.float32)
    ref4[:, :, :3] = w3
    np.testing.assert_array_equal(host(w4), ref4)
    back = zeros(3, 3, 3, 64)
    lib.vc_pad_dim_f32(stream(), P(w4), 9, 4, 3, 64, P(back))
    np.testing.assert_array_equal(host(back), w3)


@pytest.mark.parametrize("V", [11313, 1003, 6])
def test_softmax_xent_padded_pitch_keeps_the_register_kernel(lib, V):
    """V % 4 != 0 (the reference's observed vocabulary is 11313) with the row pitch padded to a multiple of 4: same values as the
    oracle, the padding columns must not influence max / sum and end up as zeros (exp(-inf)) in the in-place gradient."""
    rng = np.random.default_rng(V + 1)
    R, ld = 200, (V + 3) // 4 * 4
    logits = (rng.standard_normal((R, V), dtype=np.float32) * 3).astype(np.float32)
    padded = np.full((R, ld), 1e30, np.float32)  # poison: would dominate the max if it were read as data
    padded[:, :V] = logits
    labels = rng.integers(1, V, size=R).astype(np.int32)
    labels[::5] = 0
    loss, cache = O.xent_masked_fwd(logits.astype(np.float64), labels)
    dref = O.xent_masked_bwd(cache, 2.0)
    tl, den, rl = dev(padded), zeros(1), zeros(R)
    lib.vc_count_nonzero_i32(stream(), P(dev(labels)), R, P(den))
    lib.vc_softmax_xent_f32(stream(), P(tl), P(dev(labels)), R, V, ld, P(den), 2.0, P(rl), 1)
    got = host(tl)
    assert_close(got[:, :V], dref, 1e-5, msg="dlogits V=%d ld=%d" % (V, ld))
    assert np.all(got[:, V:] == 0)
    np.testing.assert_allclose(host(rl).sum() / host(den)[0], loss, rtol=1e-5)


@pytest.mark.parametrize("case", [(6400, 10000), (25600, 10000), (25600, 11313), (333, 50), (40, 7)], ids=lambda c: "%dx%d" % c)
def test_embedding_index_on_device_equals_the_host_index(lib, case):
    """vc_embedding_grad_index (stable counting sort on device) vs engine.embedding_grad_index (numpy argsort + bincount): same
    order, same sub-segment boundaries, same per-id ranges; unused sub-segments are empty.  Integer work: bit-exact.  Ids include
    hot tokens (<PAD> / <BOS> / <EOS> thousands of times) and out-of-range values (clipped like the gather does)."""
    from vae_captioning_amd.engine import embedding_grad_index
    R, V = case
    rng = np.random.default_rng(R + V)
    ids = rng.integers(3, V, size=R).astype(np.int32)
    ids[rng.random(R) < 0.3] = 0
    ids[rng.random(R) < 0.05] = 1
    ids[rng.random(R) < 0.05] = 2
    ids[:2] = [-5, V + 9]
    order, seg1, seg2 = embedding_grad_index(ids, V)
    nmax = int(lib.vc_embedding_index_max_subsegments(R, V, 32))
    assert nmax >= seg1.size - 1
    i32 = dict(dtype=torch.int32, device="cuda")
    d_order, d_seg1, d_seg2 = torch.full((R,), -1, **i32), torch.full((nmax + 1,), -1, **i32), torch.full((V + 1,), -1, **i32)
    ws = empty_bytes(lib.vc_embedding_index_workspace_bytes(R, V))
    for _ in range(2):  # twice: the scratch counters are rebuilt every call
        lib.vc_embedding_grad_index(stream(), P(dev(ids)), R, V, 32, P(d_order), P(d_seg1), P(d_seg2), P(ws), ws.numel() * 4)
    np.testing.assert_array_equal(host(d_order), order)
    np.testing.assert_array_equal(host(d_seg2), seg2)
    np.testing.assert_array_equal(host(d_seg1)[:seg1.size], seg1)
    assert np.all(host(d_seg1)[seg1.size:] == R)
    # and the two-level sum over the device index reproduces the dense scatter-add
    E = 8
    dX = rng.standard_normal((R, E), dtype=np.float32)
    part, table = zeros(nmax, E), zeros(V, E)
    lib.vc_embedding_grad_sorted_f32(stream(), P(part), P(d_order), P(d_seg1), E, nmax, P(dev(dX)))
    lib.vc_embedding_grad_sorted_f32(stream(), P(table), None, P(d_seg2), E, V, P(part))
    ref = np.zeros((V, E), np.float64)
    np.add.at(ref, np.clip(ids, 0, V - 1), dX.astype(np.float64))
    assert_close(host(table), ref, 1e-5, msg="embedding gradient through the device index")
