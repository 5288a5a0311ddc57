"""Patch-staged 3x3 convolution (csrc/conv_patch.hip: forward and data gradient with LDS-staged halo patches and
pre-packed weights; utils/image_embeddings.py:36-212) against the fp64 numpy oracle, through the C ABI.
Tolerance: 2e-6 * sqrt(K) of the tensor max (fp32 MFMA accumulation over K = 9*C terms vs fp64)."""
import numpy as np
import pytest
import torch

from .gpu_util import P, assert_close, dev, empty_bytes, host, stream, zeros

pytestmark = pytest.mark.gpu

from oracle import vgg as OV  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def lib():
    from vae_captioning_amd import abi
    return abi.load()


def _pack(lib, w, transpose):
    wp = torch.empty(w.numel(), dtype=torch.float32, device="cuda")
    Ci, Co = int(w.shape[2]), int(w.shape[3])
    lib.vc_conv3x3_pack_f32(stream(), Ci, Co, P(w), transpose, P(wp))
    return wp


def test_pack_layout(lib):
    """packed[tap][c/4][n][c%4]: forward = w[tap][c][n]; data gradient = w[8 - tap][n][c] (flipped taps, transposed)."""
    rng = np.random.default_rng(0)
    w = rng.standard_normal((3, 3, 8, 12), dtype=np.float32)
    tw = dev(w)
    f = host(_pack(lib, tw, 0)).reshape(9, 2, 12, 4)
    assert np.array_equal(f, w.reshape(9, 2, 4, 12).transpose(0, 1, 3, 2))
    t = host(_pack(lib, tw, 1)).reshape(9, 3, 8, 4)
    wt = w.reshape(9, 8, 12)[::-1]                       # tap' = 8 - tap
    assert np.array_equal(t, wt.reshape(9, 8, 3, 4).transpose(0, 2, 1, 3))


# (B, H, W, Cin, Cout): SUB tiling (W % 8 == 0, H % 4 == 0) incl. partial last tile, 64- and 128-wide column tiles;
# FLAT tiling (everything else with H*W >= 128): tiles that straddle two images, ragged last tile, W != H
CASES = [(2, 8, 8, 32, 64), (3, 4, 8, 32, 128), (3, 12, 16, 64, 128), (1, 56, 56, 64, 64), (2, 28, 40, 128, 256),
         (3, 14, 14, 32, 64), (2, 28, 28, 64, 128), (5, 14, 14, 96, 128), (2, 12, 20, 32, 64), (1, 20, 12, 64, 192)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_patch_fwd_dgrad_match_oracle(lib, case):
    B, H, W, Ci, Co = case
    assert lib.vc_conv3x3_patch_supported(B, H, W, Ci, Co, 0) == 1
    rng = np.random.default_rng(sum(case))
    x = np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)
    w = rng.standard_normal((3, 3, Ci, Co), dtype=np.float32) * np.float32(1 / np.sqrt(9 * Ci))
    b = rng.standard_normal(Co, dtype=np.float32)
    dy = rng.standard_normal((B, H, W, Co), dtype=np.float32)
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    pre = OV.conv3x3_fwd(x64, w64, b.astype(np.float64))
    tx, tw, tdy = dev(x), dev(w), dev(dy)
    wp = _pack(lib, tw, 0)
    y = zeros(B, H, W, Co)
    lib.vc_conv3x3_fwd_packed_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), P(dev(b)), P(y), 1, None, 0)
    assert_close(host(y), np.maximum(pre, 0), 2e-6 * np.sqrt(9 * Ci) + 1e-6, msg="patch fwd (+bias, relu)")
    lib.vc_conv3x3_fwd_packed_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), None, P(y), 0, None, 0)
    assert_close(host(y), pre - b, 2e-6 * np.sqrt(9 * Ci) + 1e-6, msg="patch fwd (no bias, no relu)")
    if lib.vc_conv3x3_patch_supported(B, H, W, Ci, Co, 1):
        dxref, _, _ = OV.conv3x3_bwd(x64, w64, dy.astype(np.float64))
        wpt = _pack(lib, tw, 1)
        dx = zeros(B, H, W, Ci)
        lib.vc_conv3x3_dgrad_packed_f32(stream(), B, H, W, Ci, Co, P(tdy), P(wpt), P(tx), P(dx), None, 0)
        assert_close(host(dx), dxref * (x > 0), 2e-6 * np.sqrt(9 * Co) + 1e-6, msg="patch dgrad (+relu mask)")
        lib.vc_conv3x3_dgrad_packed_f32(stream(), B, H, W, Ci, Co, P(tdy), P(wpt), None, P(dx), None, 0)
        assert_close(host(dx), dxref, 2e-6 * np.sqrt(9 * Co) + 1e-6, msg="patch dgrad")
    else:
        assert Ci % 64 != 0  # the data gradient's output columns are the input channels


def test_patch_unsupported_shapes_are_refused(lib):
    from vae_captioning_amd.abi import VaecapError
    assert lib.vc_conv3x3_patch_supported(2, 224, 224, 4, 64, 0) == 0     # conv1_1: 4 gathered channels
    assert lib.vc_conv3x3_patch_supported(2, 6, 6, 32, 64, 0) == 0         # FLAT needs H*W >= 128
    assert lib.vc_conv3x3_patch_supported(2, 30, 60, 32, 64, 0) == 0       # FLAT patch would not fit the LDS budget
    x, wp, y = zeros(2, 6, 6, 32), zeros(9 * 32 * 64), zeros(2, 6, 6, 64)
    with pytest.raises(VaecapError):
        lib.vc_conv3x3_fwd_packed_f32(stream(), 2, 6, 6, 32, 64, P(x), P(wp), None, P(y), 0, None, 0)


@pytest.mark.parametrize("case", [(4, 224, 224, 64, 64, "fwd"), (4, 224, 224, 64, 128, "fwd"), (3, 112, 112, 128, 128, "both"), (6, 28, 28, 512, 512, "both"),
                                  (44, 14, 14, 512, 512, "dgrad")], ids=lambda c: "x".join(map(str, c)))
def test_patch_whole_rounds_tail_split_and_old_kernel(lib, case):
    """With a workspace the tiles beyond the last whole round of resident workgroups run as a K-split launch; results
    agree with the single launch and with the implicit-GEMM kernel of conv.hip up to fp32 summation order, and are
    bit-reproducible."""
    B, H, W, Ci, Co, which = case
    rng = np.random.default_rng(B + Ci)
    x = torch.from_numpy(np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)).cuda()
    w = torch.from_numpy(rng.standard_normal((3, 3, Ci, Co), dtype=np.float32) * np.float32(1 / np.sqrt(9 * Ci))).cuda()
    b = torch.from_numpy(rng.standard_normal(Co, dtype=np.float32)).cuda()
    dy = torch.from_numpy(rng.standard_normal((B, H, W, Co), dtype=np.float32)).cuda()
    if which in ("fwd", "both"):
        wp = _pack(lib, w, 0)
        nb = lib.vc_conv3x3_packed_workspace_bytes(B, H, W, Ci, Co, 0)
        assert nb > 0, "this shape must trigger the tail split"
        ws = empty_bytes(nb)
        y0, y1, y2, y3 = (zeros(B, H, W, Co) for _ in range(4))
        lib.vc_conv3x3_fwd_packed_f32(stream(), B, H, W, Ci, Co, P(x), P(wp), P(b), P(y0), 1, None, 0)
        lib.vc_conv3x3_fwd_packed_f32(stream(), B, H, W, Ci, Co, P(x), P(wp), P(b), P(y1), 1, P(ws), ws.numel() * 4)
        lib.vc_conv3x3_fwd_packed_f32(stream(), B, H, W, Ci, Co, P(x), P(wp), P(b), P(y2), 1, P(ws), ws.numel() * 4)
        lib.vc_conv3x3_fwd_f32(stream(), B, H, W, Ci, Co, P(x), P(w), P(b), P(y3), 1, None, 0)
        assert torch.equal(y1, y2), "not bit-reproducible"
        assert_close(host(y1), host(y0), 4e-6 * np.sqrt(9 * Ci) + 1e-6, msg="fwd tail split vs single launch")
        assert_close(host(y0), host(y3), 4e-6 * np.sqrt(9 * Ci) + 1e-6, msg="patch fwd vs implicit-GEMM fwd")
    if which in ("dgrad", "both"):
        wpt = _pack(lib, w, 1)
        nb = lib.vc_conv3x3_packed_workspace_bytes(B, H, W, Ci, Co, 1)
        assert nb > 0, "this shape must trigger the tail split"
        ws = empty_bytes(nb)
        d0, d1, d3 = (zeros(B, H, W, Ci) for _ in range(3))
        for mask in (P(x), None):
            lib.vc_conv3x3_dgrad_packed_f32(stream(), B, H, W, Ci, Co, P(dy), P(wpt), mask, P(d0), None, 0)
            lib.vc_conv3x3_dgrad_packed_f32(stream(), B, H, W, Ci, Co, P(dy), P(wpt), mask, P(d1), P(ws), ws.numel() * 4)
            lib.vc_conv3x3_dgrad_f32(stream(), B, H, W, Ci, Co, P(dy), P(w), mask, P(d3), None, 0)
            assert_close(host(d1), host(d0), 4e-6 * np.sqrt(9 * Co) + 1e-6, msg="dgrad tail split vs single launch")
            assert_close(host(d0), host(d3), 4e-6 * np.sqrt(9 * Co) + 1e-6, msg="patch dgrad vs implicit-GEMM dgrad")


# ---- weight gradient ------------------------------------------------------------------------------------------------
# 4 x 8 K-tiles (W % 8 == 0, H % 4 == 0), then 4 x 7 (W % 7 == 0, H % 4 == 0: the 28-wide layers) and 2 x 14 (W % 14 == 0, H % 2 == 0:
# the 14-wide layers; also 28-wide images whose height is no multiple of 4)
WG_CASES = [(2, 8, 8, 64, 64), (3, 4, 8, 64, 128), (2, 12, 16, 128, 64), (1, 56, 56, 64, 64), (2, 28, 40, 128, 192), (5, 8, 16, 64, 64),
            (2, 28, 28, 64, 128), (1, 28, 28, 128, 256), (3, 14, 14, 128, 128), (5, 14, 14, 64, 256), (3, 20, 14, 64, 128), (2, 8, 28, 64, 128),
            (3, 6, 28, 64, 64), (2, 4, 7, 64, 64), (1, 2, 14, 64, 64)]


@pytest.mark.parametrize("case", WG_CASES, ids=lambda c: "x".join(map(str, c)))
def test_patch_wgrad_matches_oracle(lib, case):
    B, H, W, Ci, Co = case
    assert lib.vc_conv3x3_wgrad_patch_supported(B, H, W, Ci, Co) == 1
    rng = np.random.default_rng(sum(case))
    x = np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)
    w = rng.standard_normal((3, 3, Ci, Co), dtype=np.float32)
    dy = rng.standard_normal((B, H, W, Co), dtype=np.float32)
    _, dwref, dbref = OV.conv3x3_bwd(x.astype(np.float64), w.astype(np.float64), dy.astype(np.float64))
    tx, tdy = dev(x), dev(dy)
    ws = empty_bytes(lib.vc_conv3x3_wgrad_patch_workspace_bytes(B, H, W, Ci, Co))
    dw, db = zeros(3, 3, Ci, Co), zeros(Co)
    lib.vc_conv3x3_wgrad_patch_f32(stream(), B, H, W, Ci, Co, P(tx), P(tdy), P(dw), P(db), 0, P(ws), ws.numel() * 4)
    assert_close(host(dw), dwref, 2e-6 * np.sqrt(B * H * W) + 1e-6, msg="patch wgrad")
    assert_close(host(db), dbref, 1e-5, msg="patch wgrad bias gradient")
    lib.vc_conv3x3_wgrad_patch_f32(stream(), B, H, W, Ci, Co, P(tx), P(tdy), P(dw), None, 1, P(ws), ws.numel() * 4)
    assert_close(host(dw), 2 * dwref, 2e-6 * np.sqrt(B * H * W) + 1e-6, msg="patch wgrad accumulate, no bias")


def test_patch_wgrad_large_matches_implicit_gemm_and_is_reproducible(lib):
    B, H, W, Ci, Co = 8, 112, 112, 64, 128
    rng = np.random.default_rng(5)
    x = torch.from_numpy(np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)).cuda()
    dy = torch.from_numpy(rng.standard_normal((B, H, W, Co), dtype=np.float32)).cuda()
    ws = empty_bytes(max(lib.vc_conv3x3_wgrad_patch_workspace_bytes(B, H, W, Ci, Co), lib.vc_conv3x3_wgrad_workspace_bytes(B, H, W, Ci, Co)))
    d0, d1, d2 = zeros(3, 3, Ci, Co), zeros(3, 3, Ci, Co), zeros(3, 3, Ci, Co)
    b0, b2 = zeros(Co), zeros(Co)
    lib.vc_conv3x3_wgrad_patch_f32(stream(), B, H, W, Ci, Co, P(x), P(dy), P(d0), P(b0), 0, P(ws), ws.numel() * 4)
    lib.vc_conv3x3_wgrad_patch_f32(stream(), B, H, W, Ci, Co, P(x), P(dy), P(d1), None, 0, P(ws), ws.numel() * 4)
    lib.vc_conv3x3_wgrad_f32(stream(), B, H, W, Ci, Co, P(x), P(dy), P(d2), P(b2), 0, P(ws), ws.numel() * 4)
    assert torch.equal(d0, d1), "not bit-reproducible"
    assert_close(host(d0), host(d2), 4e-6 * np.sqrt(B * H * W) + 1e-6, msg="patch wgrad vs implicit-GEMM wgrad")
    assert_close(host(b0), host(b2), 1e-5, msg="bias gradient")


def test_patch_wgrad_unsupported_shapes(lib):
    assert lib.vc_conv3x3_wgrad_patch_supported(2, 28, 20, 256, 512) == 0   # W is no multiple of 8, 7 or 14
    assert lib.vc_conv3x3_wgrad_patch_supported(2, 29, 28, 256, 512) == 0   # 4 x 7 tiles need H % 4 == 0, 2 x 14 tiles H % 2 == 0
    assert lib.vc_conv3x3_wgrad_patch_supported(2, 28, 28, 256, 64) == 1    # 4 x 7 K-tiles, 64 output channels per workgroup
    assert lib.vc_conv3x3_wgrad_patch_supported(2, 224, 224, 4, 64) == 0    # conv1_1
    assert lib.vc_conv3x3_wgrad_patch_supported(2, 56, 56, 96, 64) == 0     # Cin % 64


@pytest.mark.parametrize("case", [(2, 8, 16, 32, 64), (3, 12, 8, 64, 128), (4, 224, 224, 64, 64), (3, 112, 112, 128, 128), (5, 56, 56, 256, 256),
                                  (7, 56, 56, 128, 256)], ids=lambda c: "x".join(map(str, c)))
def test_fused_maxpool_epilogue_equals_separate_pool(lib, case):
    """vc_conv3x3_fwd_pool_packed_f32 writes y and max_pool2x2(y): both bit-identical to the unfused pair (the max of the raw
    sums + bias + ReLU equals the max of the finished activations: bias add and ReLU are monotone), with and without the
    K-split tail launch (whose reduce kernel pools as well)."""
    B, H, W, Ci, Co = case
    rng = np.random.default_rng(B + H)
    x = torch.from_numpy(np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)).cuda()
    w = torch.from_numpy(rng.standard_normal((3, 3, Ci, Co), dtype=np.float32) * np.float32(1 / np.sqrt(9 * Ci))).cuda()
    b = torch.from_numpy(rng.standard_normal(Co, dtype=np.float32)).cuda()
    wp = _pack(lib, w, 0)
    nb = lib.vc_conv3x3_packed_workspace_bytes(B, H, W, Ci, Co, 0)
    ws = empty_bytes(nb)
    for use_ws in ([False, True] if nb > 0 else [False]):
        wsp, wsb = (P(ws), ws.numel() * 4) if use_ws else (None, 0)
        y0, y1 = zeros(B, H, W, Co), zeros(B, H, W, Co)
        p0, p1 = zeros(B, H // 2, W // 2, Co), torch.full((B, H // 2, W // 2, Co), -7.0, device="cuda")
        lib.vc_conv3x3_fwd_packed_f32(stream(), B, H, W, Ci, Co, P(x), P(wp), P(b), P(y0), 1, wsp, wsb)
        lib.vc_maxpool2x2_fwd_f32(stream(), B, H, W, Co, P(y0), P(p0))
        lib.vc_conv3x3_fwd_pool_packed_f32(stream(), B, H, W, Ci, Co, P(x), P(wp), P(b), P(y1), P(p1), 1, wsp, wsb)
        assert torch.equal(y0, y1)
        assert torch.equal(p0, p1), (use_ws, float((p0 - p1).abs().max()))
    from vae_captioning_amd.abi import VaecapError
    with pytest.raises(VaecapError):  # FLAT tiling: no fused pool
        lib.vc_conv3x3_fwd_pool_packed_f32(stream(), 2, 14, 14, 32, 64, P(zeros(2, 14, 14, 32)), P(zeros(9 * 32 * 64)), None, P(zeros(2, 14, 14, 64)),
                                           P(zeros(2, 7, 7, 64)), 1, None, 0)


# ----------------------------------------------------------------------------- conv1_1 (csrc/conv_first.hip)
@pytest.mark.parametrize("case", [(2, 5, 32), (1, 7, 96), (3, 16, 64), (2, 224, 224)], ids=lambda c: "x".join(map(str, c)))
def test_conv1_fwd_wgrad_match_oracle(lib, case):
    """The first layer's own kernels (3 -> 64 channels; channels as the MFMA's M dimension, 16-byte stores from the accumulators;
    weight gradient with the bias gradient as a 28th contraction row): image borders, several row segments, batch edges, the
    VGG geometry; accumulate flag; the fourth input channel must be ignored."""
    B, H, W = case
    assert lib.vc_conv1_supported(B, H, W) == 1 and lib.vc_conv1_supported(B, H, W + 8) == 0
    rng = np.random.default_rng(B + H + W)
    x = rng.standard_normal((B, H, W, 3), dtype=np.float32)
    w = rng.standard_normal((3, 3, 3, 64), dtype=np.float32) * np.float32(0.2)
    b = rng.standard_normal(64, dtype=np.float32)
    dy = rng.standard_normal((B, H, W, 64), dtype=np.float32)
    x4 = np.concatenate([x, rng.standard_normal((B, H, W, 1), dtype=np.float32)], axis=3)  # garbage in the pad channel
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    pre = OV.conv3x3_fwd(x64, w64, b.astype(np.float64))
    tx4, tw, tb, tdy = dev(x4), dev(w), dev(b), dev(dy)
    y = zeros(B, H, W, 64)
    lib.vc_conv1_fwd_f32(stream(), B, H, W, P(tx4), P(tw), P(tb), P(y), 1)
    assert_close(host(y), np.maximum(pre, 0), 2e-6 * np.sqrt(27), msg="conv1 forward + ReLU")
    lib.vc_conv1_fwd_f32(stream(), B, H, W, P(tx4), P(tw), P(tb), P(y), 0)
    assert_close(host(y), pre, 2e-6 * np.sqrt(27), msg="conv1 forward, no ReLU")
    _, dw_ref, db_ref = OV.conv3x3_bwd(x64, w64, dy.astype(np.float64), need_dx=False)
    ws = empty_bytes(lib.vc_conv1_wgrad_workspace_bytes())
    dw0 = rng.standard_normal((3, 3, 3, 64), dtype=np.float32)
    db0 = rng.standard_normal(64, dtype=np.float32)
    dw, db = dev(dw0), dev(db0)
    lib.vc_conv1_wgrad_f32(stream(), B, H, W, P(tx4), P(tdy), P(dw), P(db), 0, P(ws), ws.numel() * 4)
    tol = 2e-6 * np.sqrt(B * H * W)
    assert_close(host(dw), dw_ref, tol, msg="conv1 weight gradient")
    assert_close(host(db), db_ref, tol, msg="conv1 bias gradient")
    first = host(dw).copy()
    lib.vc_conv1_wgrad_f32(stream(), B, H, W, P(tx4), P(tdy), P(dw), P(db), 0, P(ws), ws.numel() * 4)
    assert np.array_equal(host(dw), first), "weight gradient is not bit-reproducible"
    dw, db = dev(dw0), dev(db0)
    lib.vc_conv1_wgrad_f32(stream(), B, H, W, P(tx4), P(tdy), P(dw), P(db), 1, P(ws), ws.numel() * 4)
    assert_close(host(dw), dw_ref + dw0, tol, msg="conv1 weight gradient, accumulate")
    assert_close(host(db), db_ref + db0, tol, msg="conv1 bias gradient, accumulate")
    from vae_captioning_amd.abi import VaecapError
    with pytest.raises(VaecapError):
        lib.vc_conv1_wgrad_f32(stream(), B, H, W, P(tx4), P(tdy), P(dw), P(db), 0, None, 0)


# ----------------------------------------------------------------------------- MaxPoolGrad fused into the data gradient
@pytest.mark.parametrize("case", [(2, 8, 8, 64, 64), (3, 12, 16, 64, 128), (1, 56, 56, 64, 128), (2, 28, 28, 128, 256), (3, 14, 14, 128, 64),
                                  (2, 112, 112, 64, 128)], ids=lambda c: "x".join(map(str, c)))
def test_dgrad_with_fused_unpool_equals_dgrad_then_maxpool_bwd(lib, case):
    """vc_conv3x3_dgrad_unpool_packed_f32 == vc_maxpool2x2_bwd_f32(y, vc_conv3x3_dgrad_packed_f32(dy), relu_grad = 1) bit for bit:
    both tilings, both tile widths, with and without the K-split tail launch; ties and non-positive maxima occur in y."""
    B, H, W, Ci, Co = case            # the convolution behind the pool: input [B,H,W,Ci] = the pooled tensor
    rng = np.random.default_rng(B * H + Co)
    w = torch.from_numpy(rng.standard_normal((3, 3, Ci, Co), dtype=np.float32) * np.float32(1 / np.sqrt(9 * Co))).cuda()
    dy = torch.from_numpy(rng.standard_normal((B, H, W, Co), dtype=np.float32)).cuda()
    ypre = np.maximum(rng.standard_normal((B, 2 * H, 2 * W, Ci), dtype=np.float32), 0)   # ReLU output: many zeros, hence ties
    ypre[0, :4, :4] = 1.5                                                                 # a constant patch: four-way ties
    ypre = torch.from_numpy(ypre).cuda()
    wpt = _pack(lib, w, 1)
    nb = lib.vc_conv3x3_packed_workspace_bytes(B, H, W, Ci, Co, 1)
    ws = empty_bytes(nb)
    for use_ws in ([False, True] if nb > 0 else [False]):
        wsp, wsb = (P(ws), ws.numel() * 4) if use_ws else (None, 0)
        dpool = zeros(B, H, W, Ci)
        ref, got = zeros(B, 2 * H, 2 * W, Ci), torch.full((B, 2 * H, 2 * W, Ci), 7.0, device="cuda")
        lib.vc_conv3x3_dgrad_packed_f32(stream(), B, H, W, Ci, Co, P(dy), P(wpt), None, P(dpool), wsp, wsb)
        lib.vc_maxpool2x2_bwd_f32(stream(), B, 2 * H, 2 * W, Ci, P(ypre), P(dpool), P(ref), 1)
        lib.vc_conv3x3_dgrad_unpool_packed_f32(stream(), B, H, W, Ci, Co, P(dy), P(wpt), P(ypre), P(got), wsp, wsb)
        assert torch.equal(ref, got), "fused unpool differs (workspace %s)" % use_ws
