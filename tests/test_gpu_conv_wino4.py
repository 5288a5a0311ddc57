"""Winograd F(4x4, 3x3) convolution (csrc/conv_wino4.hip: forward and data gradient of utils/image_embeddings.py:36-212 with 36
positions per 4 x 4 output tile, blocks of 4 x 4 tiles) against the fp64 numpy oracle, through the C ABI.  Activations go in and come
out in the C4 layout (include/vaecap.h); the oracle is NHWC, the tests convert on the host (tests/gpu_util.py to_c4 / from_c4).
Tolerance: 6e-5 of the tensor max, FLAT (no sqrt(K)) -- the transforms multiply by 4, 5, 8 and 1/24, which costs about a decimal
digit against F(2x2,3x3) (held to 3e-6 * sqrt(K)); measured on the VGG16 shapes: 1e-5 of the tensor maximum."""
import numpy as np
import pytest
import torch

from .gpu_util import P, assert_close, dev, dev_c4, host, host_c4, stream, zeros

pytestmark = pytest.mark.gpu

from oracle import vgg as OV  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def lib():
    from vae_captioning_amd import abi
    return abi.load()


def _pack(lib, w, transpose):
    wp = torch.empty(36 * w.shape[2] * w.shape[3], dtype=torch.float32, device="cuda")
    lib.vc_conv3x3_wino4_pack_f32(stream(), int(w.shape[2]), int(w.shape[3]), P(w), transpose, P(wp))
    return wp


def test_wino4_pack_is_G_g_Gt(lib):
    rng = np.random.default_rng(0)
    Ci, Co = 32, 64
    w = rng.standard_normal((3, 3, Ci, Co), dtype=np.float32)
    G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
    for transpose in (0, 1):
        g = w.astype(np.float64) if not transpose else w[::-1, ::-1].transpose(0, 1, 3, 2).astype(np.float64)
        C, N = g.shape[2], g.shape[3]
        V = np.einsum("uk,klcn,vl->vucn", G, g, G).reshape(36, C, N)              # position p = 6 v + u (u: vertical index)
        # packed [nt][phase][group][pq][g][n][pp]: channel = 4 phase + g, column = 32 nt + 16 group + n, position = 4 pq + pp
        got = host(_pack(lib, dev(w), transpose)).reshape(N // 32, C // 4, 2, 9, 4, 16, 4)
        ref = V.reshape(9, 4, C // 4, 4, N // 32, 2, 16).transpose(4, 2, 5, 0, 3, 6, 1)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)


# (B, H, W, Cin, Cout): whole blocks (16, 32, 48 wide), ragged blocks (28, 14, 20, 12: tiles and pixels past the image), tiny images,
# W != H, a ragged last workgroup (odd block count), several column tiles, the shortest reduction (two phases)
CASES = [(2, 16, 16, 32, 64), (3, 4, 8, 8, 32), (1, 56, 56, 64, 64), (2, 28, 28, 64, 128), (3, 14, 14, 32, 64), (5, 14, 14, 96, 128),
         (2, 12, 20, 32, 64), (1, 20, 12, 64, 96), (1, 32, 48, 16, 32), (1, 6, 10, 48, 32), (1, 112, 112, 16, 32)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_wino4_fwd_dgrad_match_oracle(lib, case):
    B, H, W, Ci, Co = case
    assert lib.vc_conv3x3_wino4_supported(B, H, W, Ci, Co, 0) == 1
    rng = np.random.default_rng(sum(case))
    x = np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)
    w = rng.standard_normal((3, 3, Ci, Co), dtype=np.float32) * np.float32(1 / np.sqrt(9 * Ci))
    b = rng.standard_normal(Co, dtype=np.float32)
    dy = rng.standard_normal((B, H, W, Co), dtype=np.float32)
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    if B * H * W <= 20000:
        pre = OV.conv3x3_fwd(x64, w64, b.astype(np.float64))
        dxref = OV.conv3x3_bwd(x64, w64, dy.astype(np.float64))[0] if Ci % 32 == 0 and Co % 8 == 0 else None
    else:   # the 112 x 112 case: torch fp64 on the device as the reference of the reference (same contraction)
        t = lambda a: torch.from_numpy(a).cuda().double()
        wt = t(w).permute(3, 2, 0, 1)
        pre = torch.nn.functional.conv2d(t(x).permute(0, 3, 1, 2), wt, t(b), padding=1).permute(0, 2, 3, 1).cpu().numpy()
        dxref = torch.nn.grad.conv2d_input((B, Ci, H, W), wt, t(dy).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).cpu().numpy() if Ci % 32 == 0 else None
    tx, tw, tdy = dev_c4(x), dev(w), dev_c4(dy)
    wp = _pack(lib, tw, 0)
    y = zeros(B, H, W, Co)
    ys, ps = (B, H, W, Co), (B, H // 2, W // 2, Co)
    tol = 6e-5
    lib.vc_conv3x3_wino4_fwd_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), P(dev(b)), P(y), None, 1)
    assert_close(host_c4(y, ys), np.maximum(pre, 0), tol, msg="wino4 fwd (+bias, relu)")
    lib.vc_conv3x3_wino4_fwd_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), None, P(y), None, 0)
    assert_close(host_c4(y, ys), pre - b, tol, msg="wino4 fwd (no bias, no relu)")
    yp = zeros(B, H // 2, W // 2, Co)
    y.zero_()
    lib.vc_conv3x3_wino4_fwd_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), P(dev(b)), P(y), P(yp), 1)
    hy = host_c4(y, ys)
    assert_close(hy, np.maximum(pre, 0), tol, msg="wino4 fwd + pool: y")
    assert np.array_equal(host_c4(yp, ps), hy.reshape(B, H // 2, 2, W // 2, 2, Co).max(axis=(2, 4))), "fused pool != max_pool2x2(y)"
    if dxref is not None:
        assert lib.vc_conv3x3_wino4_supported(B, H, W, Ci, Co, 1) == 1
        wpt = _pack(lib, tw, 1)
        dx = zeros(B, H, W, Ci)
        told, xs = 6e-5, (B, H, W, Ci)
        lib.vc_conv3x3_wino4_dgrad_f32(stream(), B, H, W, Ci, Co, P(tdy), P(wpt), P(tx), P(dx))
        assert_close(host_c4(dx, xs), dxref * (x > 0), told, msg="wino4 dgrad (+relu mask)")
        lib.vc_conv3x3_wino4_dgrad_f32(stream(), B, H, W, Ci, Co, P(tdy), P(wpt), None, P(dx))
        assert_close(host_c4(dx, xs), dxref, told, msg="wino4 dgrad")


def test_wino4_error_is_a_digit_above_f23_and_far_inside_the_tolerance(lib):
    """What the constants of F(4x4,3x3) cost, measured where it is used: conv4_2's reduction (K = 9 * 512), post-ReLU inputs, He-scaled
    weights -- the error against fp64 is ~1e-5 of the tensor maximum (F(2x2,3x3): ~1e-6), the test tolerance 6e-5 (flat)."""
    B, H, Ci, Co = 2, 28, 512, 64
    rng = np.random.default_rng(5)
    x = np.maximum(rng.standard_normal((B, H, H, Ci), dtype=np.float32), 0)
    w = rng.standard_normal((3, 3, Ci, Co), dtype=np.float32) * np.float32(np.sqrt(2.0 / (9 * Ci)))
    pre = OV.conv3x3_fwd(x.astype(np.float64), w.astype(np.float64), np.zeros(Co))
    tx, tw = dev_c4(x), dev(w)
    y4, y2 = zeros(B, H, H, Co), zeros(B, H, H, Co)
    lib.vc_conv3x3_wino4_fwd_f32(stream(), B, H, H, Ci, Co, P(tx), P(_pack(lib, tw, 0)), None, P(y4), None, 0)
    wp2 = torch.empty(16 * Ci * Co, dtype=torch.float32, device="cuda")
    lib.vc_conv3x3_wino_pack_f32(stream(), Ci, Co, P(tw), 0, P(wp2))
    lib.vc_conv3x3_wino_fwd_f32(stream(), B, H, H, Ci, Co, P(tx), P(wp2), None, P(y2), None, 0)
    scale = np.abs(pre).max()
    e4, e2 = np.abs(host_c4(y4, pre.shape) - pre).max() / scale, np.abs(host_c4(y2, pre.shape) - pre).max() / scale
    assert e2 < 2e-6 and e4 < 4e-5 and e4 < 40 * max(e2, 1e-7), (e2, e4)


def test_wino4_unsupported_shapes_are_refused_and_preference_follows_the_block_coverage(lib):
    from vae_captioning_amd.abi import VaecapError
    assert lib.vc_conv3x3_wino4_supported(2, 224, 224, 4, 64, 0) == 0     # conv1_1: 4 gathered channels
    assert lib.vc_conv3x3_wino4_supported(2, 8, 8, 32, 48, 0) == 0         # output channels % 32
    assert lib.vc_conv3x3_wino4_supported(2, 8, 8, 32, 64, 1) == 1 and lib.vc_conv3x3_wino4_supported(2, 8, 8, 24, 64, 1) == 0
    x, wp, y = zeros(2, 8, 8, 32), zeros(36 * 32 * 48), zeros(2, 8, 8, 48)
    with pytest.raises(VaecapError):
        lib.vc_conv3x3_wino4_fwd_f32(stream(), 2, 8, 8, 32, 48, P(x), P(wp), None, P(y), None, 0)
    # VGG16 at 224 x 224: every 3x3 layer behind conv1_1 since round 4.  With linear tiles (blocks of sixteen consecutive tiles: the 56-, 28-
    # and 40-wide images below) only the padding inside the 4 x 4 tiles of the right / bottom edge is left over: 14 x 14 -> 77 % of the slots,
    # above the 0.75 x F(2x2,3x3)'s coverage the rule asks for; odd sizes stay with the other kernels (no fused pool)
    pref = {H: lib.vc_conv3x3_wino4_preferred(32, H, H, 64, 64) for H in (224, 112, 56, 28, 14, 40, 27)}
    assert pref == {224: 1, 112: 1, 56: 1, 28: 1, 14: 1, 40: 1, 27: 0}, pref
    assert lib.vc_conv3x3_wino4_preferred(32, 224, 224, 4, 64) == 0


@pytest.mark.parametrize("case", [(2, 16, 16, 32, 64), (1, 56, 56, 64, 64), (2, 28, 28, 64, 128), (3, 14, 14, 32, 64), (2, 12, 20, 32, 64)], ids=lambda c: "x".join(map(str, c)))
def test_wino4_relu_mask_as_bits(lib, case):
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
    nw = lib.vc_conv3x3_wino4_mask_words(B, H, W, Cm)
    assert nw > 0
    bits = torch.zeros(nw, dtype=torch.int32, device="cuda")
    wp1 = _pack(lib, w1, 0)
    lib.vc_conv3x3_wino4_fwd_f32(stream(), B, H, W, Ci, Cm, P(x), P(wp1), P(b1), P(y0), None, 1)
    lib.vc_conv3x3_wino4_fwd_mask_f32(stream(), B, H, W, Ci, Cm, P(x), P(wp1), P(b1), P(y), 1, P(bits))
    assert np.array_equal(host(y), host(y0))
    wpt2 = _pack(lib, w2, 1)
    dx_ref, dx = zeros(B, H, W, Cm), zeros(B, H, W, Cm)
    lib.vc_conv3x3_wino4_dgrad_f32(stream(), B, H, W, Cm, Co, P(dy), P(wpt2), P(y), P(dx_ref))
    lib.vc_conv3x3_wino4_dgrad_bits_f32(stream(), B, H, W, Cm, Co, P(dy), P(wpt2), P(bits), P(dx))
    assert np.array_equal(host(dx), host(dx_ref))
    assert float(np.abs(host(dx_ref)).max()) > 0 and (host(dx_ref) == 0).mean() > 0.2   # the mask does mask


@pytest.mark.parametrize("case", [(2, 16, 16, 32, 64), (1, 56, 56, 64, 64), (3, 28, 28, 32, 96), (5, 14, 14, 64, 32), (2, 12, 20, 16, 64)], ids=lambda c: "x".join(map(str, c)))
def test_wino4_pool_routing_codes_equal_maxpool_bwd_on_the_activation(lib, case):
    """The pooled forward also leaves MaxPoolGrad's routing codes in conv_wino.hip's format (a lane owns four channels = one half-word);
    routing the pooled gradient with them is bit-identical to vc_maxpool2x2_bwd_f32 on the pre-pool activation, ties included."""
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
    lib.vc_conv3x3_wino4_fwd_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), P(tb), P(y0), P(p0), 1)
    y1, p1 = zeros(B, H, W, Co), zeros(B, H // 2, W // 2, Co)
    nw = lib.vc_conv3x3_wino_pool_words(B, H, W, Co)
    bits = torch.zeros(nw, dtype=torch.int32, device="cuda")
    lib.vc_conv3x3_wino4_fwd_pool_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), P(tb), P(y1), P(p1), P(bits))
    assert torch.equal(y0, y1) and torch.equal(p0, p1)
    dy = dev_c4(rng.standard_normal((B, H // 2, W // 2, Co), dtype=np.float32))
    d_ref, d_bits = zeros(B, H, W, Co), torch.full((B, H, W, Co), 7.0, device="cuda")
    lib.vc_maxpool2x2_bwd_f32(stream(), B * (Co // 4), H, W, 4, P(y1), P(dy), P(d_ref), 1)   # (C4 planes of four-channel pixels)
    lib.vc_maxpool2x2_bwd_bits_f32(stream(), B, H, W, Co, P(bits), P(dy), P(d_bits))
    assert torch.equal(d_ref, d_bits)
    hy = host_c4(y1, (B, H, W, Co))
    win = hy.reshape(B, H // 2, 2, W // 2, 2, Co)
    assert (win.max(axis=(2, 4)) == 0).mean() > 0.05 and ((win == win.max(axis=(2, 4), keepdims=True)).sum(axis=(2, 4)) > 1).mean() > 0.1   # ties do occur


def test_wino4_calls_over_the_launch_limit_are_cut_into_image_ranges(tmp_path):
    """Like conv_wino.hip, the kernel addresses a launch's tensors with 32-bit offsets; a call on more images runs as launches over image
    ranges (cfg4's 512-image step on one GPU does).  VC_WINO_MAX_BYTES forces that on a small shape (a fresh process: the limit is read
    once): forward, pooled forward with routing codes and data gradient must equal the single-launch results bit for bit."""
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
        "B, H, W, Ci, Co = 5, 12, 20, 64, 64\n"
        "g = torch.Generator(device='cuda').manual_seed(1)\n"
        "x = torch.rand(B, H, W, Ci, device='cuda', generator=g); w = torch.rand(3, 3, Ci, Co, device='cuda', generator=g) - 0.5\n"
        "dy = torch.rand(B, H, W, Co, device='cuda', generator=g) - 0.5; b = torch.rand(Co, device='cuda', generator=g) - 0.5\n"
        "vp = torch.empty(36 * Ci * Co, device='cuda'); vpt = torch.empty(36 * Ci * Co, device='cuda')\n"
        "lib.vc_conv3x3_wino4_pack_f32(st, Ci, Co, P(w), 0, P(vp)); lib.vc_conv3x3_wino4_pack_f32(st, Ci, Co, P(w), 1, P(vpt))\n"
        "y = torch.zeros(B, H, W, Co, device='cuda'); yp = torch.zeros(B, H // 2, W // 2, Co, device='cuda'); dx = torch.zeros(B, H, W, Ci, device='cuda')\n"
        "y2 = torch.zeros_like(y); yp2 = torch.zeros_like(yp); bits = torch.zeros(lib.vc_conv3x3_wino_pool_words(B, H, W, Co), dtype=torch.int32, device='cuda')\n"
        "lib.vc_conv3x3_wino4_fwd_f32(st, B, H, W, Ci, Co, P(x), P(vp), P(b), P(y), P(yp), 1)\n"
        "lib.vc_conv3x3_wino4_fwd_pool_f32(st, B, H, W, Ci, Co, P(x), P(vp), P(b), P(y2), P(yp2), P(bits))\n"
        "lib.vc_conv3x3_wino4_dgrad_f32(st, B, H, W, Ci, Co, P(dy), P(vpt), P(x), P(dx))\n"
        "torch.cuda.synchronize()\n"
        "print(int(lib.vc_conv3x3_wino_single_launch_supported(B, H, W, Ci, Co)))\n"
        "np.savez(sys.argv[1], y=y.cpu().numpy(), yp=yp.cpu().numpy(), y2=y2.cpu().numpy(), yp2=yp2.cpu().numpy(), bits=bits.cpu().numpy(), dx=dx.cpu().numpy())\n" % root)
    out = {}
    for tag, cap in (("one", None), ("cut", str(2 * 12 * 20 * 64 * 4))):   # cut: two images per launch -> ranges of 2, 2, 1
        env = dict(os.environ)
        env.pop("VC_WINO_MAX_BYTES", None)
        if cap:
            env["VC_WINO_MAX_BYTES"] = cap
        f = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, str(script), f], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.strip().splitlines()[-1] == ("1" if cap is None else "0")
        out[tag] = np.load(f)
    for k in ("y", "yp", "y2", "yp2", "bits", "dx"):
        assert np.array_equal(out["one"][k], out["cut"][k]), k
    assert np.array_equal(out["one"]["y"], out["one"]["y2"]) and np.abs(out["one"]["dx"]).max() > 0 and out["one"]["bits"].any()


@pytest.mark.parametrize("case", [(3, 32, 64), (2, 224, 224), (1, 16, 32)], ids=lambda c: "x".join(map(str, c)))
def test_conv1_forward_leaves_the_mask_bits_of_conv1_2s_data_gradient(lib, case):
    """vc_conv1_fwd_mask_f32: conv1_1's forward (its own kernel: a lane = one pixel x eight channel quads) also writes (y > 0) as bits in
    the lane order of the F(4x4,3x3) data gradient with 64 produced channels (a lane = a 4 x 4-pixel tile x one channel quad): the same
    y as vc_conv1_fwd_f32, and vc_conv3x3_wino4_dgrad_bits_f32 on those bits equals vc_conv3x3_wino4_dgrad_f32 on the float activation
    bit for bit (utils/image_embeddings.py:36-63: ReluGrad of conv1_1's output inside conv1_2's backward)."""
    B, H, W = case
    g = torch.Generator(device="cuda").manual_seed(B * H + W)
    x4 = torch.rand(B, H, W, 4, device="cuda", generator=g) * 255 - 120
    x4[..., 3] = 0
    w = (torch.rand(3, 3, 3, 64, device="cuda", generator=g) - 0.5) * 0.02
    b = torch.rand(64, device="cuda", generator=g) - 0.5
    y1, y2 = zeros(B, 16, H, W, 4), zeros(B, 16, H, W, 4)
    bits = torch.full((lib.vc_conv3x3_wino4_mask_words(B, H, W, 64),), -1, dtype=torch.int32, device="cuda")   # (every word must be written)
    lib.vc_conv1_fwd_f32(stream(), B, H, W, P(x4), P(w), P(b), P(y1), 1)
    lib.vc_conv1_fwd_mask_f32(stream(), B, H, W, P(x4), P(w), P(b), P(y2), P(bits))
    assert torch.equal(y1, y2) and 0.2 < float((y1 > 0).float().mean()) < 0.8
    w2 = (torch.rand(3, 3, 64, 64, device="cuda", generator=g) - 0.5) * 0.1
    wpt = torch.empty(36 * 64 * 64, device="cuda")
    lib.vc_conv3x3_wino4_pack_f32(stream(), 64, 64, P(w2), 1, P(wpt))
    dy = torch.rand(B, 16, H, W, 4, device="cuda", generator=g) - 0.5
    dxa, dxb = zeros(B, 16, H, W, 4), zeros(B, 16, H, W, 4)
    lib.vc_conv3x3_wino4_dgrad_f32(stream(), B, H, W, 64, 64, P(dy), P(wpt), P(y1), P(dxa))
    lib.vc_conv3x3_wino4_dgrad_bits_f32(stream(), B, H, W, 64, 64, P(dy), P(wpt), P(bits), P(dxb))
    assert torch.equal(dxa, dxb) and float(dxa.abs().max()) > 0
    with pytest.raises(Exception):
        lib.vc_conv1_fwd_mask_f32(stream(), B, H + 8, W, P(x4), P(w), P(b), P(y2), P(bits))   # H % 16 != 0 is refused


# ---- round 6: the same convolution on a once-transformed input (csrc/conv_wino4.hip MODE 2, vc_conv3x3_wino4v_*) -------------------------------
# shapes whose launch puts a tile in the fused kernel's lane: linear-tile layers (28 / 56 wide, incl. ragged last blocks, an odd block
# count, one image, tile counts that are no multiple of 16) and 13..16-pixel images (one square block per image); several channel tiles,
# the shortest reduction
V_CASES = [(2, 28, 28, 64, 128), (3, 28, 28, 32, 32), (3, 28, 28, 8, 64), (5, 14, 14, 96, 128), (3, 14, 14, 32, 64), (1, 56, 56, 64, 64),
           (2, 16, 16, 32, 64), (2, 13, 15, 16, 32), (7, 28, 28, 256, 64), (2, 8, 8, 32, 32)]


@pytest.mark.parametrize("case", V_CASES, ids=lambda c: "x".join(map(str, c)))
def test_wino4v_is_bit_identical_to_the_fused_kernel_and_matches_the_oracle(lib, case):
    """utils/image_embeddings.py:96-212 (the layers behind pool2) forward + tf.gradients' data gradient: the pre-transformed form runs the
    same 1-D transforms in the same order and the same MFMA sequence as the fused kernel, so EVERYTHING it leaves -- activation, ReLU mask
    bits, pooled activation, MaxPoolGrad routing codes, data gradient with the mask as bits and as floats -- must equal the fused
    entry's output bit for bit; the forward is also held to the fp64 oracle (6e-5 flat, the F(4x4,3x3) bound)."""
    B, H, W, Ci, Co = case
    assert lib.vc_conv3x3_wino4v_supported(B, H, W, Ci, Co, 0) == 1
    rng = np.random.default_rng(sum(case) + 1)
    x = np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)
    w = rng.standard_normal((3, 3, Ci, Co), dtype=np.float32) * np.float32(1 / np.sqrt(9 * Ci))
    b = rng.standard_normal(Co, dtype=np.float32)
    dy = rng.standard_normal((B, H, W, Co), dtype=np.float32)
    tx, tw, tb, tdy = dev_c4(x), dev(w), dev(b), dev_c4(dy)
    wp = _pack(lib, tw, 0)
    nbytes = max(lib.vc_conv3x3_wino4v_workspace_bytes(B, H, W, Ci), lib.vc_conv3x3_wino4v_workspace_bytes(B, H, W, Co))
    assert lib.vc_conv3x3_wino4v_workspace_bytes(B, H, W, Ci) == -(-(B * -(-H // 4) * -(-W // 4)) // 16) * Ci * 2304
    vws = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device="cuda")   # (stale NaNs: the transform must write every slot it reads)
    mw = lib.vc_conv3x3_wino4_mask_words(B, H, W, Co)
    y, mk = [zeros(B, H, W, Co) for _ in range(2)], [torch.zeros(mw, dtype=torch.int32, device="cuda") for _ in range(2)]
    lib.vc_conv3x3_wino4_fwd_mask_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), P(tb), P(y[0]), 1, P(mk[0]))
    lib.vc_conv3x3_wino4v_fwd_mask_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), P(tb), P(y[1]), 1, P(mk[1]), P(vws), nbytes)
    assert torch.equal(y[0], y[1]) and torch.equal(mk[0], mk[1])
    if B * H * W <= 20000:
        pre = OV.conv3x3_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
        assert_close(host_c4(y[1], (B, H, W, Co)), np.maximum(pre, 0), 6e-5, msg="wino4v fwd (+bias, relu)")
    lib.vc_conv3x3_wino4_fwd_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), None, P(y[0]), None, 0)
    lib.vc_conv3x3_wino4v_fwd_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), None, P(y[1]), None, 0, P(vws), nbytes)
    assert torch.equal(y[0], y[1])
    if H % 2 == 0 and W % 2 == 0:
        yp = [zeros(B, H // 2, W // 2, Co) for _ in range(2)]
        pb = [torch.zeros(lib.vc_conv3x3_wino_pool_words(B, H, W, Co), dtype=torch.int32, device="cuda") for _ in range(2)]
        lib.vc_conv3x3_wino4_fwd_pool_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), P(tb), P(y[0]), P(yp[0]), P(pb[0]))
        lib.vc_conv3x3_wino4v_fwd_pool_f32(stream(), B, H, W, Ci, Co, P(tx), P(wp), P(tb), P(y[1]), P(yp[1]), P(pb[1]), P(vws), nbytes)
        assert torch.equal(y[0], y[1]) and torch.equal(yp[0], yp[1]) and torch.equal(pb[0], pb[1])
    if Ci % 32 == 0 and Co % 8 == 0 and lib.vc_conv3x3_wino4v_supported(B, H, W, Ci, Co, 1):
        wpt = _pack(lib, tw, 1)
        dx = [zeros(B, H, W, Ci) for _ in range(2)]
        mi = torch.zeros(lib.vc_conv3x3_wino4_mask_words(B, H, W, Ci), dtype=torch.int32, device="cuda").random_(0, 2 ** 31 - 1)
        vws.fill_(float("nan"))
        lib.vc_conv3x3_wino4_dgrad_bits_f32(stream(), B, H, W, Ci, Co, P(tdy), P(wpt), P(mi), P(dx[0]))
        lib.vc_conv3x3_wino4v_dgrad_bits_f32(stream(), B, H, W, Ci, Co, P(tdy), P(wpt), P(mi), P(dx[1]), P(vws), nbytes)
        assert torch.equal(dx[0], dx[1])
        lib.vc_conv3x3_wino4_dgrad_f32(stream(), B, H, W, Ci, Co, P(tdy), P(wpt), P(tx), P(dx[0]))
        lib.vc_conv3x3_wino4v_dgrad_f32(stream(), B, H, W, Ci, Co, P(tdy), P(wpt), P(tx), P(dx[1]), P(vws), nbytes)
        assert torch.equal(dx[0], dx[1])
        if B * H * W <= 20000:
            dxref = OV.conv3x3_bwd(x.astype(np.float64), w.astype(np.float64), dy.astype(np.float64))[0] * (x > 0)
            assert_close(host_c4(dx[1], (B, H, W, Ci)), dxref, 6e-5, msg="wino4v dgrad (relu mask from the activation)")


def test_wino4v_refuses_what_it_does_not_take(lib):
    """square-block layers whose lanes differ from the linear order (224 / 112 wide, 32 wide), a workspace that is too small or missing"""
    from vae_captioning_amd.abi import VaecapError
    assert lib.vc_conv3x3_wino4v_supported(2, 224, 224, 64, 64, 0) == 0 and lib.vc_conv3x3_wino4v_supported(2, 32, 32, 32, 32, 0) == 0
    assert lib.vc_conv3x3_wino4v_supported(2, 8, 8, 32, 32, 0) == 1          # (two 8 x 8 images fill ONE linear block: the fused kernel takes linear blocks there too)
    assert lib.vc_conv3x3_wino4v_supported(1, 8, 8, 32, 32, 0) == 0          # one image = one square block of the fused kernel, lanes (ty, tx) != the linear order at 2 tiles per row
    assert lib.vc_conv3x3_wino4v_supported(2, 28, 28, 30, 32, 0) == 0         # channels
    assert lib.vc_conv3x3_wino4v_workspace_bytes(2, 224, 224, 64) == 0
    # the measured preference rule: the 28- and 14-wide layers with >= 256 gathered channels, not the 56-wide ones
    assert lib.vc_conv3x3_wino4v_preferred(32, 28, 28, 512, 512, 0) == 1 and lib.vc_conv3x3_wino4v_preferred(32, 14, 14, 512, 512, 1) == 1
    assert lib.vc_conv3x3_wino4v_preferred(32, 28, 28, 256, 512, 0) == 1 and lib.vc_conv3x3_wino4v_preferred(32, 56, 56, 256, 256, 0) == 0
    assert lib.vc_conv3x3_wino4v_preferred(32, 28, 28, 128, 512, 0) == 0
    B, H, W, Ci, Co = 1, 28, 28, 32, 32
    x, wp, y = zeros(B, H, W, Ci), torch.zeros(36 * Ci * Co, device="cuda"), zeros(B, H, W, Co)
    need = lib.vc_conv3x3_wino4v_workspace_bytes(B, H, W, Ci)
    vws = torch.zeros(need // 4, device="cuda")
    with pytest.raises(VaecapError):
        lib.vc_conv3x3_wino4v_fwd_f32(stream(), B, H, W, Ci, Co, P(x), P(wp), None, P(y), None, 0, P(vws), need - 4)
    with pytest.raises(VaecapError):
        lib.vc_conv3x3_wino4v_fwd_f32(stream(), B, H, W, Ci, Co, P(x), P(wp), None, P(y), None, 0, None, need)
    with pytest.raises(VaecapError):
        lib.vc_conv3x3_wino4v_fwd_f32(stream(), 2, 32, 32, Ci, Co, P(x), P(wp), None, P(y), None, 0, P(vws), need)
