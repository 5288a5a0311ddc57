"""Winograd F(2x2, 3x3) convolution (csrc/conv_wino.hip: forward and data gradient of utils/image_embeddings.py:36-212 with the
input / output transforms fused around sixteen MFMA position products on 16-tile blocks, weights transformed once per step; weight
gradient F(3x3, 2x2): csrc/conv_wino_wgrad.hip) against the fp64 numpy
oracle, through the C ABI.  Tolerance: 3e-6 * sqrt(K) of the tensor max, K = 9*C (the direct kernels are held to 2e-6 * sqrt(K);
the Winograd transforms add a few more fp32 roundings per term).  Activations go in and come out in the C4 layout [B][C/4][H][W][4]
(include/vaecap.h); the oracle is NHWC, the tests convert on the host (tests/gpu_util.py to_c4 / from_c4)."""
import numpy as np
import pytest
import torch

from .gpu_util import P, assert_close, dev, dev_c4, empty_bytes, from_c4, host, host_c4, stream, to_c4, zeros

pytestmark = pytest.mark.gpu

from oracle import vgg as OV  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def lib():
    from vae_captioning_amd import abi
    return abi.load()


def _pack(lib, w, transpose):
    wp = torch.empty(16 * w.shape[2] * w.shape[3], dtype=torch.float32, device="cuda")
    lib.vc_conv3x3_wino_pack_f32(stream(), int(w.shape[2]), int(w.shape[3]), P(w), transpose, P(wp))
    return wp


@pytest.mark.parametrize("shape", [(2, 6, 10, 8), (1, 7, 5, 64), (3, 4, 4, 4), (2, 14, 14, 512)], ids=lambda c: "x".join(map(str, c)))
def test_layout_conversion_kernels_match_numpy(lib, shape):
    """vc_nhwc_to_c4_f32 / vc_c4_to_nhwc_f32 (the fc1 boundary of the trainer, and what the -m gpu tests use on device tensors) against
    the numpy transposition; C = 4 is the identity (conv1_1's zero-padded RGB input is both layouts at once)."""
    B, H, W, C = shape
    x = np.random.default_rng(C).standard_normal(shape, dtype=np.float32)
    out = zeros(B, C // 4, H, W, 4)
    lib.vc_nhwc_to_c4_f32(stream(), B, H, W, C, P(dev(x)), P(out))
    assert np.array_equal(host(out), to_c4(x))
    back = zeros(*shape)
    lib.vc_c4_to_nhwc_f32(stream(), B, H, W, C, P(out), P(back))
    assert np.array_equal(host(back), x) and np.array_equal(from_c4(to_c4(x)), x)
    if C == 4:
        assert np.array_equal(host(out).reshape(shape), x)


def test_wino_pack_is_G_g_Gt(lib):
    rng = np.random.default_rng(0)
    Ci, Co = 32, 64
    w = rng.standard_normal((3, 3, Ci, Co), dtype=np.float32)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
    for transpose in (0, 1):
        g = w.astype(np.float64) if not transpose else w[::-1, ::-1].transpose(0, 1, 3, 2).astype(np.float64)
        C, N = g.shape[2], g.shape[3]
        V = np.einsum("ak,klcn,bl->abcn", G, g, G).reshape(16, C, N)                        # [p][c][n]
        # packed [nt][chunk][half][p][g][n][ct][e], channel = 16 chunk + 8 half + 2 g + e, column = 32 nt + 8 (n >> 2) + 4 ct + (n & 3)
        got = host(_pack(lib, dev(w), transpose)).reshape(N // 32, C // 16, 2, 16, 4, 4, 4, 2, 2)       # n split as (n >> 2, n & 3)
        ref = V.reshape(16, C // 16, 2, 4, 2, N // 32, 4, 2, 4).transpose(5, 1, 2, 0, 3, 6, 8, 7, 4)      # column as (nt, n >> 2, ct, n & 3)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)


# (B, H, W, Cin, Cout): block shapes 4x4 (224 / 112 / 56 wide), 2x7 (28, 14: ragged rows), other fits (3x5, 2x4 ...), tiny images,
# W != H, a ragged last workgroup (block count not a multiple of 4), several column tiles
CASES = [(2, 8, 8, 32, 64), (3, 4, 8, 16, 32), (1, 56, 56, 64, 64), (2, 28, 28, 64, 128), (3, 14, 14, 32, 64), (5, 14, 14, 96, 128),
         (2, 12, 20, 32, 64), (1, 20, 12, 64, 96), (1, 32, 48, 16, 32), (2, 2, 2, 16, 32), (1, 6, 10, 48, 32)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_wino_fwd_dgrad_match_oracle(lib, case):
    B, H, W, Ci, Co = case
    assert lib.vc_conv3x3_wino_supported(B, H, W, Ci, Co, 0) == 1
    rng = np.random.default_rng(sum(case))
    x = np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)
    w = rng.standard_normal((3, 3, Ci, Co), dtype=np.float32) * np.float32(1 / np.sqrt(9 * Ci))
    b = rng.standard_normal(Co, dtype=np.float32)
    dy = rng.standard_normal((B, H, W, Co), dtype=np.float32)
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    pre = OV.conv3x3_fwd(x64, w64, b.astype(np.float64))
    tx, tw, tdy = dev_c4(x), dev(w), dev_c4(dy)
    wp = _pack(lib, tw, 0)
    y = zeros(B, H, W, Co)
    ys, ps, xs = (B, H, W, Co), (B, H // 2, W // 2, Co), (B, H, W, Ci)
    tol = 3e-6 * np.sqrt(9 * Ci) + 1e-6
    lib.vc_conv3x3_wino_fwd_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), P(dev(b)), P(y), None, 1)
    assert_close(host_c4(y, ys), np.maximum(pre, 0), tol, msg="wino fwd (+bias, relu)")
    lib.vc_conv3x3_wino_fwd_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), None, P(y), None, 0)
    assert_close(host_c4(y, ys), pre - b, tol, msg="wino fwd (no bias, no relu)")
    yp = zeros(B, H // 2, W // 2, Co)
    y.zero_()
    lib.vc_conv3x3_wino_fwd_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), P(dev(b)), P(y), P(yp), 1)
    hy = host_c4(y, ys)
    assert_close(hy, np.maximum(pre, 0), tol, msg="wino fwd + pool: y")
    assert np.array_equal(host_c4(yp, ps), hy.reshape(B, H // 2, 2, W // 2, 2, Co).max(axis=(2, 4))), "fused pool != max_pool2x2(y)"
    if lib.vc_conv3x3_wino_supported(B, H, W, Ci, Co, 1):
        dxref, _, _ = OV.conv3x3_bwd(x64, w64, dy.astype(np.float64))
        wpt = _pack(lib, tw, 1)
        dx = zeros(B, H, W, Ci)
        told = 3e-6 * np.sqrt(9 * Co) + 1e-6
        lib.vc_conv3x3_wino_dgrad_f32(stream(), B, H, W, Ci, Co, P(tdy), P(wpt), P(tx), P(dx))
        assert_close(host_c4(dx, xs), dxref * (x > 0), told, msg="wino dgrad (+relu mask)")
        lib.vc_conv3x3_wino_dgrad_f32(stream(), B, H, W, Ci, Co, P(tdy), P(wpt), None, P(dx))
        assert_close(host_c4(dx, xs), dxref, told, msg="wino dgrad")
    else:
        assert Ci % 32 != 0 or Co % 16 != 0


def test_wino_unsupported_shapes_are_refused(lib):
    from vae_captioning_amd.abi import VaecapError
    assert lib.vc_conv3x3_wino_supported(2, 224, 224, 4, 64, 0) == 0     # conv1_1: 4 gathered channels
    assert lib.vc_conv3x3_wino_supported(2, 7, 8, 32, 64, 0) == 0         # odd height
    assert lib.vc_conv3x3_wino_supported(2, 8, 8, 32, 48, 0) == 0         # output channels % 32
    x, wp, y = zeros(2, 7, 8, 32), zeros(16 * 32 * 64), zeros(2, 7, 8, 64)
    with pytest.raises(VaecapError):
        lib.vc_conv3x3_wino_fwd_f32(stream(), 2, 7, 8, 32, 64, P(x), P(wp), None, P(y), None, 0)


# (B, H, W, Cin, Cout): block shapes 4x8 / 4x7 / 2x14, ragged blocks at the image border (H, W not multiples of the block), a ragged
# last K split, several channel blocks, more workgroups than blocks per split
WG_CASES = [(2, 8, 16, 64, 64), (1, 56, 56, 64, 64), (2, 28, 28, 64, 128), (3, 14, 14, 128, 64), (2, 12, 20, 64, 64), (5, 4, 6, 64, 192), (1, 224, 224, 64, 64)]


@pytest.mark.parametrize("case", WG_CASES, ids=lambda c: "x".join(map(str, c)))
def test_wino_wgrad_matches_oracle(lib, case):
    B, H, W, Ci, Co = case
    assert lib.vc_conv3x3_wino_wgrad_supported(B, H, W, Ci, Co) == 1
    rng = np.random.default_rng(sum(case) + 1)
    x = np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)
    dy = rng.standard_normal((B, H, W, Co), dtype=np.float32)
    w0 = np.zeros((3, 3, Ci, Co))
    if B * H * W <= 20000:
        _, dwref, dbref = OV.conv3x3_bwd(x.astype(np.float64), w0, dy.astype(np.float64))
    else:   # the 224 x 224 case: torch fp64 on the device as the reference of the reference (same contraction)
        tx64, tdy64 = torch.from_numpy(x).cuda().double().permute(0, 3, 1, 2), torch.from_numpy(dy).cuda().double().permute(0, 3, 1, 2)
        dwref = torch.nn.grad.conv2d_weight(tx64, (Co, Ci, 3, 3), tdy64, padding=1).permute(2, 3, 1, 0).cpu().numpy()
        dbref = dy.astype(np.float64).sum(axis=(0, 1, 2))
    ws = empty_bytes(lib.vc_conv3x3_wino_wgrad_workspace_bytes(B, H, W, Ci, Co))
    dw, db = zeros(3, 3, Ci, Co), zeros(Co)
    lib.vc_conv3x3_wino_wgrad_f32(stream(), B, H, W, Ci, Co, P(dev_c4(x)), P(dev_c4(dy)), P(dw), P(db), 0, P(ws), ws.numel() * 4)
    tol = 3e-6 * np.sqrt(B * H * W) + 1e-6
    assert_close(host(dw), dwref, tol, msg="wino wgrad")
    assert_close(host(db), dbref, tol, msg="wino wgrad: bias gradient")
    first = host(dw).copy()
    lib.vc_conv3x3_wino_wgrad_f32(stream(), B, H, W, Ci, Co, P(dev_c4(x)), P(dev_c4(dy)), P(dw), P(db), 1, P(ws), ws.numel() * 4)
    assert_close(host(dw), 2 * dwref, tol, msg="wino wgrad (accumulate)")
    assert_close(host(db), 2 * dbref, tol, msg="wino wgrad (accumulate): bias gradient")
    dw2 = zeros(3, 3, Ci, Co)
    lib.vc_conv3x3_wino_wgrad_f32(stream(), B, H, W, Ci, Co, P(dev_c4(x)), P(dev_c4(dy)), P(dw2), None, 0, P(ws), ws.numel() * 4)
    assert np.array_equal(host(dw2), first), "not bit-reproducible"


@pytest.mark.parametrize("case", [(2, 8, 8, 32, 64), (1, 56, 56, 64, 64), (2, 28, 28, 64, 128), (3, 14, 14, 32, 64), (2, 12, 20, 32, 64)], ids=lambda c: "x".join(map(str, c)))
def test_wino_relu_mask_as_bits(lib, case):
    """forward of layer L leaves (y > 0) as bits; the data gradient of layer L + 1 (Cin = L's Cout, same H x W) reads them: bit-identical
    to the data gradient that loads relu_src = y, and y itself is what the plain forward writes"""
    B, H, W, Ci, Cm = case
    Co = 64
    rng = np.random.default_rng(sum(case) + 3)
    x = dev(rng.standard_normal((B, H, W, Ci), dtype=np.float32))
    w1 = dev(rng.standard_normal((3, 3, Ci, Cm), dtype=np.float32) * np.float32(0.1))
    b1 = dev(rng.standard_normal(Cm, dtype=np.float32))
    w2 = dev(rng.standard_normal((3, 3, Cm, Co), dtype=np.float32) * np.float32(0.1))
    dy = dev(rng.standard_normal((B, H, W, Co), dtype=np.float32))
    y, y0 = zeros(B, H, W, Cm), zeros(B, H, W, Cm)
    nw = lib.vc_conv3x3_wino_mask_words(B, H, W, Cm)
    assert nw > 0
    bits = torch.zeros(nw, dtype=torch.int32, device="cuda")
    wp1 = _pack(lib, w1, 0)
    lib.vc_conv3x3_wino_fwd_f32(stream(), B, H, W, Ci, Cm, P(x), P(wp1), P(b1), P(y0), None, 1)
    lib.vc_conv3x3_wino_fwd_mask_f32(stream(), B, H, W, Ci, Cm, P(x), P(wp1), P(b1), P(y), 1, P(bits))
    assert np.array_equal(host(y), host(y0))
    wpt2 = _pack(lib, w2, 1)
    dx_ref, dx = zeros(B, H, W, Cm), zeros(B, H, W, Cm)
    lib.vc_conv3x3_wino_dgrad_f32(stream(), B, H, W, Cm, Co, P(dy), P(wpt2), P(y), P(dx_ref))
    lib.vc_conv3x3_wino_dgrad_bits_f32(stream(), B, H, W, Cm, Co, P(dy), P(wpt2), P(bits), P(dx))
    assert np.array_equal(host(dx), host(dx_ref))
    assert float(np.abs(host(dx_ref)).max()) > 0 and (host(dx_ref) == 0).mean() > 0.2   # the mask does mask


def test_wino_calls_over_the_launch_limit_are_cut_into_image_ranges(tmp_path):
    """The kernels address a launch's tensors with 32-bit offsets (< 2 GiB); a call on more images runs as several launches over image
    ranges.  VC_WINO_MAX_BYTES forces that on a small shape (a fresh process: the limit is read once): forward / data gradient must
    equal the single-launch results bit for bit, the weight gradient up to the summation order of the ranges."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from vae_captioning_amd import abi\n"
        "from vae_captioning_amd.abi import ptr as P\n"
        "lib = abi.load(); st = torch.cuda.current_stream().cuda_stream\n"
        "B, H, W, Ci, Co = 5, 8, 16, 64, 64\n"
        "g = torch.Generator(device='cuda').manual_seed(1)\n"
        "x = torch.rand(B, H, W, Ci, device='cuda', generator=g); w = torch.rand(3, 3, Ci, Co, device='cuda', generator=g) - 0.5\n"
        "dy = torch.rand(B, H, W, Co, device='cuda', generator=g) - 0.5; b = torch.rand(Co, device='cuda', generator=g)\n"
        "vp = torch.empty(16 * Ci * Co, device='cuda'); vpt = torch.empty(16 * Ci * Co, device='cuda')\n"
        "lib.vc_conv3x3_wino_pack_f32(st, Ci, Co, P(w), 0, P(vp)); lib.vc_conv3x3_wino_pack_f32(st, Ci, Co, P(w), 1, P(vpt))\n"
        "y = torch.zeros(B, H, W, Co, device='cuda'); yp = torch.zeros(B, H // 2, W // 2, Co, device='cuda'); dx = torch.zeros(B, H, W, Ci, device='cuda')\n"
        "dw = torch.zeros(3, 3, Ci, Co, device='cuda'); db = torch.zeros(Co, device='cuda')\n"
        "ws = torch.empty(lib.vc_conv3x3_wino_wgrad_workspace_bytes(B, H, W, Ci, Co) // 4 + 4, device='cuda')\n"
        "lib.vc_conv3x3_wino_fwd_f32(st, B, H, W, Ci, Co, P(x), P(vp), P(b), P(y), P(yp), 1)\n"
        "lib.vc_conv3x3_wino_dgrad_f32(st, B, H, W, Ci, Co, P(dy), P(vpt), P(x), P(dx))\n"
        "lib.vc_conv3x3_wino_wgrad_f32(st, B, H, W, Ci, Co, P(x), P(dy), P(dw), P(db), 0, P(ws), ws.numel() * 4)\n"
        "torch.cuda.synchronize()\n"
        "print(int(lib.vc_conv3x3_wino_single_launch_supported(B, H, W, Ci, Co)))\n"
        "np.savez(sys.argv[1], y=y.cpu().numpy(), yp=yp.cpu().numpy(), dx=dx.cpu().numpy(), dw=dw.cpu().numpy(), db=db.cpu().numpy())\n" % root)
    out = {}
    for tag, cap in (("one", None), ("cut", str(2 * 8 * 16 * 64 * 4))):   # cut: two images per launch -> ranges of 2, 2, 1
        env = dict(os.environ)
        env.pop("VC_WINO_MAX_BYTES", None)
        if cap:
            env["VC_WINO_MAX_BYTES"] = cap
        f = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, str(script), f], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.strip().splitlines()[-1] == ("1" if cap is None else "0")
        out[tag] = np.load(f)
    for k in ("y", "yp", "dx"):
        assert np.array_equal(out["one"][k], out["cut"][k]), k
    for k in ("dw", "db"):
        assert np.abs(out["one"][k] - out["cut"][k]).max() <= 1e-5 * np.abs(out["one"][k]).max(), k


@pytest.mark.parametrize("case", [(2, 8, 8, 32, 64), (1, 56, 56, 64, 64), (3, 28, 28, 32, 96), (5, 14, 14, 64, 32), (2, 12, 20, 16, 64)], ids=lambda c: "x".join(map(str, c)))
def test_pool_routing_codes_equal_maxpool_bwd_on_the_activation(lib, case):
    """The pooled forward also leaves MaxPoolGrad's routing codes (first maximum of the 2x2 window | valid if > 0); routing the pooled
    gradient with them is bit-identical to vc_maxpool2x2_bwd_f32 on the pre-pool activation (ties -- whole windows of zeros behind the
    ReLU, equal positives -- included), and y / ypool equal the plain pooled forward."""
    B, H, W, Ci, Co = case
    rng = np.random.default_rng(sum(case) + 11)
    x = np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)
    x[:, : H // 2] = np.round(x[:, : H // 2])                      # small integers: exact ties between window positions
    w = np.round(rng.standard_normal((3, 3, Ci, Co), dtype=np.float32))
    w[:, :, :, : Co // 4] = 0                                       # whole channels at the bias value: four-way ties
    b = np.concatenate([np.full(Co // 8, -1.0), np.full(Co // 8, 2.0), rng.standard_normal(Co - Co // 4)]).astype(np.float32)
    tx, tw, tb = dev_c4(x), dev(w), dev(b)
    wp = _pack(lib, tw, 0)
    y0, p0 = zeros(B, H, W, Co), zeros(B, H // 2, W // 2, Co)
    lib.vc_conv3x3_wino_fwd_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), P(tb), P(y0), P(p0), 1)
    y1, p1 = zeros(B, H, W, Co), zeros(B, H // 2, W // 2, Co)
    nw = lib.vc_conv3x3_wino_pool_words(B, H, W, Co)
    assert nw == B * (H // 2) * (W // 2) * (Co // 8)
    bits = torch.zeros(nw, dtype=torch.int32, device="cuda")
    lib.vc_conv3x3_wino_fwd_pool_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), P(tb), P(y1), P(p1), P(bits))
    assert torch.equal(y0, y1) and torch.equal(p0, p1)
    dy = dev_c4(rng.standard_normal((B, H // 2, W // 2, Co), dtype=np.float32))
    d_ref, d_bits = zeros(B, H, W, Co), torch.full((B, H, W, Co), 7.0, device="cuda")
    lib.vc_maxpool2x2_bwd_f32(stream(), B * (Co // 4), H, W, 4, P(y1), P(dy), P(d_ref), 1)   # (C4 planes of four-channel pixels)
    lib.vc_maxpool2x2_bwd_bits_f32(stream(), B, H, W, Co, P(bits), P(dy), P(d_bits))
    assert torch.equal(d_ref, d_bits)
    hy = host_c4(y1, (B, H, W, Co))
    win = hy.reshape(B, H // 2, 2, W // 2, 2, Co)
    assert (win.max(axis=(2, 4)) == 0).mean() > 0.05 and ((win == win.max(axis=(2, 4), keepdims=True)).sum(axis=(2, 4)) > 1).mean() > 0.1   # ties do occur
