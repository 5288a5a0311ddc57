"""-m gpu: the PROTOTYPE f32 Winograd F(4x4,3x3) kernel on the one-wave-per-SIMD recipe (csrc/conv_wino4r.hip) -- forward and data
gradient of utils/image_embeddings.py:36-212 against the fp64 numpy oracle (blocks of 32 x 32 pixels that stick out of the image, one
to several eight-channel steps, several channel tiles) at the F(4x4,3x3) path's tolerance, and the VGG16 layer shapes at a 32-image
launch against the independent f32 implicit-GEMM kernels."""
import numpy as np
import pytest
import torch

from oracle import vgg as OV
from .gpu_util import P, assert_close, dev, dev_c4, empty_bytes, host_c4, stream, zeros

pytestmark = pytest.mark.gpu
TOL = 6e-5


def _pack(lib, w, transpose):
    from . import gpu_util
    ci, co = int(w.shape[2]), int(w.shape[3])
    wp = torch.empty(lib.vc_conv3x3_wino4r_pack_bytes(ci, co) // 4, dtype=torch.float32, device="cuda")
    lib.vc_conv3x3_wino4r_pack_f32(stream(), ci, co, P(w), transpose, P(wp))
    gpu_util._KEEP.append(wp)   # alive until the end of the test: the library holds only the raw pointer (see gpu_util.dev)
    return wp


SHAPES = [(2, 16, 32, 32, 64), (2, 12, 16, 64, 64), (3, 8, 8, 16, 128), (2, 28, 28, 64, 128), (3, 14, 14, 64, 256), (1, 9, 37, 32, 128), (5, 6, 24, 96, 128),
          (2, 20, 40, 64, 64), (1, 1, 1, 8, 32), (2, 35, 3, 48, 64), (1, 70, 100, 16, 64)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_wino4r_forward_and_data_gradient_match_the_fp64_oracle(lib, shape):
    B, H, W, Ci, Co = shape
    rng = np.random.default_rng(sum(shape))
    x = np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)       # a post-ReLU activation
    w = (rng.standard_normal((3, 3, Ci, Co), dtype=np.float32) / np.sqrt(9 * Ci)).astype(np.float32)
    b = rng.standard_normal(Co, dtype=np.float32)
    dy = rng.standard_normal((B, H, W, Co), dtype=np.float32)
    x64, w64, dy64 = x.astype(np.float64), w.astype(np.float64), dy.astype(np.float64)
    assert lib.vc_conv3x3_wino4r_supported(B, H, W, Ci, Co, 0) == 1
    wd = dev(w)
    # forward + bias + ReLU
    y_ref = np.maximum(OV.conv3x3_fwd(x64, w64, b.astype(np.float64)), 0)
    y = zeros(B, Co // 4, H, W, 4)
    lib.vc_conv3x3_wino4r_fwd_f32(stream(), B, H, W, Ci, Co, P(dev_c4(x)), P(_pack(lib, wd, 0)), P(dev(b)), P(y), 1)
    got = host_c4(y, (B, H, W, Co))
    assert_close(got, y_ref, TOL, msg="wino4r forward %s" % (shape,))
    # data gradient + ReLU mask (needs the produced channels -- this layer's INPUT channels -- to be a multiple of 32)
    if lib.vc_conv3x3_wino4r_supported(B, H, W, Ci, Co, 1):
        dx_ref, _, _ = OV.conv3x3_bwd(x64, w64, dy64)
        dx_ref = dx_ref * (x > 0)
        dx = zeros(B, Ci // 4, H, W, 4)
        lib.vc_conv3x3_wino4r_dgrad_f32(stream(), B, H, W, Ci, Co, P(dev_c4(dy)), P(_pack(lib, wd, 1)), P(dev_c4(x)), P(dx))
        assert_close(host_c4(dx, (B, H, W, Ci)), dx_ref, TOL, msg="wino4r data gradient %s" % (shape,))
        dx2 = zeros(B, Ci // 4, H, W, 4)
        lib.vc_conv3x3_wino4r_dgrad_f32(stream(), B, H, W, Ci, Co, P(dev_c4(dy)), P(_pack(lib, wd, 1)), None, P(dx2))
        assert_close(host_c4(dx2, (B, H, W, Ci)), OV.conv3x3_bwd(x64, w64, dy64)[0], TOL, msg="wino4r data gradient, no mask %s" % (shape,))
    else:
        assert Ci % 32


def test_wino4r_rejects_what_it_cannot_tile(lib):
    from vae_captioning_amd.abi import VaecapError
    assert lib.vc_conv3x3_wino4r_supported(2, 8, 8, 12, 64, 0) == 0     # contracted channels not a multiple of 8
    assert lib.vc_conv3x3_wino4r_supported(2, 8, 8, 32, 48, 0) == 0     # produced channels not a multiple of 32
    x, y = zeros(2, 8, 8, 8, 4), zeros(2, 12, 8, 8, 4)
    with pytest.raises(VaecapError):
        lib.vc_conv3x3_wino4r_fwd_f32(stream(), 2, 8, 8, 32, 48, P(x), P(x), None, P(y), 0)


LAYERS = [("conv1_2", 224, 64, 64), ("conv2_1", 112, 64, 128), ("conv2_2", 112, 128, 128), ("conv3_1", 56, 128, 256),
          ("conv3_2", 56, 256, 256), ("conv4_1", 28, 256, 512), ("conv4_2", 28, 512, 512), ("conv5_2", 14, 512, 512)]


def _c4(lib, t):
    B, H, W, C = (int(v) for v in t.shape)
    out = torch.empty(B, C // 4, H, W, 4, dtype=torch.float32, device="cuda")
    lib.vc_nhwc_to_c4_f32(stream(), B, H, W, C, P(t), P(out))
    return out


def _nhwc(lib, t, shape):
    B, H, W, C = shape
    out = torch.empty(B, H, W, C, dtype=torch.float32, device="cuda")
    lib.vc_c4_to_nhwc_f32(stream(), B, H, W, C, P(t), P(out))
    return out


@pytest.mark.parametrize("layer", LAYERS, ids=lambda l: l[0])
def test_wino4r_vgg_layer_at_the_bench_launch_size_matches_the_f32_implicit_gemm(lib, layer):
    name, H, Ci, Co = layer
    B, W = 32, H
    g = torch.Generator(device="cuda").manual_seed(H + Ci)
    x = torch.rand(B, H, W, Ci, device="cuda", generator=g).sub_(0.4).clamp_(min=0)
    w = (torch.rand(3, 3, Ci, Co, device="cuda", generator=g) - 0.5) * float(2.0 / np.sqrt(9 * Ci))
    b = torch.rand(Co, device="cuda", generator=g) - 0.5
    dy = torch.rand(B, H, W, Co, device="cuda", generator=g) - 0.5
    ws = empty_bytes(max(lib.vc_conv3x3_fwd_workspace_bytes(B, H, W, Ci, Co), lib.vc_conv3x3_dgrad_workspace_bytes(B, H, W, Ci, Co)))
    st = stream()
    xc, dyc = _c4(lib, x), _c4(lib, dy)
    y_ref, y = zeros(B, H, W, Co), zeros(B, Co // 4, H, W, 4)
    lib.vc_conv3x3_fwd_f32(st, B, H, W, Ci, Co, P(x), P(w), P(b), P(y_ref), 1, P(ws), ws.numel() * 4)
    lib.vc_conv3x3_wino4r_fwd_f32(st, B, H, W, Ci, Co, P(xc), P(_pack(lib, w, 0)), P(b), P(y), 1)
    got = _nhwc(lib, y, (B, H, W, Co))
    scale = float(y_ref.abs().max())
    assert float((got - y_ref).abs().max()) <= TOL * scale, (name, float((got - y_ref).abs().max()), scale)
    dx_ref, dx = zeros(B, H, W, Ci), zeros(B, Ci // 4, H, W, 4)
    lib.vc_conv3x3_dgrad_f32(st, B, H, W, Ci, Co, P(dy), P(w), P(x), P(dx_ref), P(ws), ws.numel() * 4)
    lib.vc_conv3x3_wino4r_dgrad_f32(st, B, H, W, Ci, Co, P(dyc), P(_pack(lib, w, 1)), P(xc), P(dx))
    gotd = _nhwc(lib, dx, (B, H, W, Ci))
    scale = float(dx_ref.abs().max())
    assert float((gotd - dx_ref).abs().max()) <= TOL * scale, (name, float((gotd - dx_ref).abs().max()), scale)
