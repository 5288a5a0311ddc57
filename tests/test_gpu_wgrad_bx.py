"""vc_conv3x3_bx_wgrad_f32 (csrc/conv_wgrad_bx.hip): the direct 3x3 weight gradient on the bf16 matrix pipe against the fp64 oracle
(oracle/vgg.py: conv3x3_bwd restates the backward of tf.nn.conv2d, utils/image_embeddings.py:36-212) -- small shapes that exercise every
edge of the block scheme (image boundaries inside a block of eight stacked rows, widths that are not multiples of sixteen, one-row and
one-column images, K splits shorter than the staging pipeline), and the VGG16 layer shapes at the bench's launch size against the f32
Winograd kernel and a torch fp64 contraction on the device."""
import numpy as np
import pytest
import torch

from oracle import vgg as OV
from .gpu_util import P, assert_close, dev_c4, empty_bytes, host, stream, zeros

pytestmark = pytest.mark.gpu

# split-bf16 products: each a.b is good to ~2^-17 relative, the errors are zero-mean; a sum of K products of N(0,1) x relu(N(0,1)) values
# has magnitude ~sqrt(K) -- the tolerance is 4x the f32 Winograd kernel's (tests/test_gpu_conv_wino.py), relative to max|dw|
def _tol(B, H, W):
    return 1.2e-5 * np.sqrt(B * H * W) + 4e-6


def _run(lib, x, dy, want_db=True, accumulate=0, dw=None, db=None):
    B, H, W, Ci = x.shape
    Co = dy.shape[3]
    ws = empty_bytes(lib.vc_conv3x3_bx_wgrad_workspace_bytes(B, H, W, Ci, Co))
    ws.fill_(float("nan"))   # every workspace element the reduce reads must have been written by the kernel
    dw = zeros(3, 3, Ci, Co) if dw is None else dw
    db = (zeros(Co) if db is None else db) if want_db else None
    lib.vc_conv3x3_bx_wgrad_f32(stream(), B, H, W, Ci, Co, P(dev_c4(x)), P(dev_c4(dy)), P(dw), P(db) if want_db else None, accumulate, P(ws), ws.numel() * 4)
    return dw, db


CASES = [(2, 8, 8, 64, 64), (1, 56, 56, 64, 64), (2, 28, 28, 64, 128), (3, 14, 14, 128, 64), (2, 12, 20, 64, 64), (5, 7, 16, 64, 64),
         (1, 1, 1, 64, 64), (3, 1, 37, 64, 64), (2, 33, 1, 64, 64), (1, 9, 17, 64, 192), (4, 3, 5, 192, 64), (1, 112, 48, 64, 64)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_bx_wgrad_matches_oracle(lib, case):
    B, H, W, Ci, Co = case
    assert lib.vc_conv3x3_bx_wgrad_supported(B, H, W, Ci, Co) == 1
    rng = np.random.default_rng(sum(case) + 11)
    x = np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)
    dy = rng.standard_normal((B, H, W, Co), dtype=np.float32)
    _, dwref, dbref = OV.conv3x3_bwd(x.astype(np.float64), np.zeros((3, 3, Ci, Co)), dy.astype(np.float64))
    dw, db = _run(lib, x, dy)
    tol = _tol(B, H, W)
    assert_close(host(dw), dwref, tol, msg="bx wgrad")
    assert_close(host(db), dbref, 3e-6 * np.sqrt(B * H * W) + 1e-6, msg="bx wgrad: bias gradient (f32 sums)")
    first = host(dw).copy()
    _run(lib, x, dy, accumulate=1, dw=dw, db=db)
    assert_close(host(dw), 2 * dwref, tol, msg="bx wgrad (accumulate)")
    assert_close(host(db), 2 * dbref, 3e-6 * np.sqrt(B * H * W) + 1e-6, msg="bx wgrad (accumulate): bias gradient")
    dw2, _ = _run(lib, x, dy, want_db=False)
    assert np.array_equal(host(dw2), first), "not bit-reproducible"


def test_bx_wgrad_error_is_split_bf16_not_bf16(lib):
    """the result must sit ~2^-16 from the fp64 value, not 2^-8 (a dropped lo term would still pass a loose tolerance): unit-scale inputs
    with a common sign so that nothing cancels"""
    B, H, W, Ci, Co = 2, 16, 16, 64, 64
    rng = np.random.default_rng(5)
    x = (1.0 + rng.random((B, H, W, Ci), dtype=np.float32)).astype(np.float32)
    dy = (1.0 + rng.random((B, H, W, Co), dtype=np.float32)).astype(np.float32)
    _, dwref, _ = OV.conv3x3_bwd(x.astype(np.float64), np.zeros((3, 3, Ci, Co)), dy.astype(np.float64))
    dw, _ = _run(lib, x, dy)
    rel = np.abs(host(dw) - dwref) / np.abs(dwref)
    assert rel.max() < 2e-5, rel.max()


def test_bx_wgrad_many_images_of_a_small_layer(lib):
    """forty images of a 6 x 10 layer: 35 blocks of stacked rows, most of them spanning two images, more K splits than blocks per split"""
    B, H, W, Ci, Co = 40, 6, 10, 64, 64
    rng = np.random.default_rng(9)
    x = np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)
    dy = rng.standard_normal((B, H, W, Co), dtype=np.float32)
    _, dwref, dbref = OV.conv3x3_bwd(x.astype(np.float64), np.zeros((3, 3, Ci, Co)), dy.astype(np.float64))
    dw, db = _run(lib, x, dy)
    assert_close(host(dw), dwref, _tol(B, H, W), msg="bx wgrad, 40 images")
    assert_close(host(db), dbref, 3e-6 * np.sqrt(B * H * W) + 1e-6, msg="bx wgrad, 40 images: bias gradient")


def test_bx_wgrad_calls_over_the_launch_limit_are_cut_into_image_ranges(tmp_path):
    """32-bit buffer offsets (< 2 GiB per launch): a call on more images runs as launches over image ranges that accumulate into dw / db.
    VC_WINO_MAX_BYTES forces that on a small shape (a fresh process: the limit is read once); equal to the single launch up to the
    summation order of the ranges, and to the f32 Winograd kernel's result on the same cut."""
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
        "x = torch.rand(B, Ci // 4, H, W, 4, device='cuda', generator=g); dy = torch.rand(B, Co // 4, H, W, 4, device='cuda', generator=g) - 0.5\n"
        "out = {}\n"
        "for name in ('bx', 'wino'):\n"
        "    dw = torch.zeros(3, 3, Ci, Co, device='cuda'); db = torch.zeros(Co, device='cuda')\n"
        "    ws = torch.empty(getattr(lib, 'vc_conv3x3_%%s_wgrad_workspace_bytes' %% name)(B, H, W, Ci, Co) // 4 + 4, device='cuda')\n"
        "    getattr(lib, 'vc_conv3x3_%%s_wgrad_f32' %% name)(st, B, H, W, Ci, Co, P(x), P(dy), P(dw), P(db), 0, P(ws), ws.numel() * 4)\n"
        "    torch.cuda.synchronize()\n"
        "    out['dw_' + name] = dw.cpu().numpy(); out['db_' + name] = db.cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n" % root)
    out = {}
    for tag, cap in (("one", None), ("cut", str(2 * 8 * 16 * 64 * 4))):   # cut: two images per launch -> ranges of 2, 2, 1
        env = dict(os.environ)
        env.pop("VC_WINO_MAX_BYTES", None)
        if cap:
            env["VC_WINO_MAX_BYTES"] = cap
        f = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, str(script), f], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = np.load(f)
    scale = np.abs(out["one"]["dw_wino"]).max()
    assert np.abs(out["one"]["dw_bx"] - out["cut"]["dw_bx"]).max() <= 1e-5 * scale
    assert np.abs(out["one"]["db_bx"] - out["cut"]["db_bx"]).max() <= 1e-5 * np.abs(out["one"]["db_bx"]).max()
    assert np.abs(out["cut"]["dw_bx"] - out["cut"]["dw_wino"]).max() <= 5e-5 * scale


def test_bx_wgrad_rejects_what_it_cannot_tile(lib):
    assert lib.vc_conv3x3_bx_wgrad_supported(2, 8, 8, 32, 64) == 0
    assert lib.vc_conv3x3_bx_wgrad_supported(2, 8, 8, 64, 96) == 0
    assert lib.vc_conv3x3_bx_wgrad_supported(0, 8, 8, 64, 64) == 0
    assert lib.vc_conv3x3_bx_wgrad_workspace_bytes(2, 8, 8, 32, 64) == 0
    from vae_captioning_amd.abi import VaecapError
    x, dy, dw = zeros(2 * 8 * 8 * 32), zeros(2 * 8 * 8 * 64), zeros(9 * 32 * 64)
    with pytest.raises(VaecapError):
        lib.vc_conv3x3_bx_wgrad_f32(stream(), 2, 8, 8, 32, 64, P(x), P(dy), P(dw), None, 0, P(x), 16)
    ws = empty_bytes(lib.vc_conv3x3_bx_wgrad_workspace_bytes(2, 8, 8, 64, 64))
    x = zeros(2 * 8 * 8 * 64)
    dw = zeros(9 * 64 * 64)
    with pytest.raises(VaecapError):   # workspace too small
        lib.vc_conv3x3_bx_wgrad_f32(stream(), 2, 8, 8, 64, 64, P(x), P(dy), P(dw), None, 0, P(ws), 1024)


VGG_LAYERS = [("conv1_2", 224, 64, 64), ("conv2_1", 112, 64, 128), ("conv2_2", 112, 128, 128), ("conv3_1", 56, 128, 256), ("conv3_2", 56, 256, 256),
              ("conv4_1", 28, 256, 512), ("conv4_2", 28, 512, 512), ("conv5_2", 14, 512, 512)]


@pytest.mark.parametrize("layer", VGG_LAYERS, ids=lambda l: l[0])
def test_bx_wgrad_vgg_layer_at_the_bench_launch_size(lib, layer):
    """32 images (the launch size of the 64-image step's two streams): against a torch fp64 contraction of the same tensors on the device,
    and within the split-bf16 error of the f32 Winograd kernel the trainer uses in f32 mode"""
    name, H, Ci, Co = layer
    B = 32 if H < 224 else 8
    g = torch.Generator(device="cuda").manual_seed(H + Ci)
    x = torch.randn(B, H, H, Ci, device="cuda", generator=g).clamp_(min=0)
    dy = torch.randn(B, H, H, Co, device="cuda", generator=g)
    xc = x.view(B, H, H, Ci // 4, 4).permute(0, 3, 1, 2, 4).contiguous()
    dyc = dy.view(B, H, H, Co // 4, 4).permute(0, 3, 1, 2, 4).contiguous()
    dwref = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2), (Co, Ci, 3, 3), dy.double().permute(0, 3, 1, 2), padding=1).permute(2, 3, 1, 0)
    dbref = dy.double().sum(dim=(0, 1, 2))
    ws = empty_bytes(max(lib.vc_conv3x3_bx_wgrad_workspace_bytes(B, H, H, Ci, Co), lib.vc_conv3x3_wino_wgrad_workspace_bytes(B, H, H, Ci, Co)))
    dw, db = zeros(3, 3, Ci, Co), zeros(Co)
    lib.vc_conv3x3_bx_wgrad_f32(stream(), B, H, H, Ci, Co, P(xc), P(dyc), P(dw), P(db), 0, P(ws), ws.numel() * 4)
    torch.cuda.synchronize()
    tol = _tol(B, H, H)
    assert_close(host(dw), dwref.cpu().numpy(), tol, msg="%s bx wgrad vs fp64" % name)
    assert_close(host(db), dbref.cpu().numpy(), 3e-6 * np.sqrt(B * H * H) + 1e-6, msg="%s bx wgrad bias" % name)
    dwf = zeros(3, 3, Ci, Co)
    lib.vc_conv3x3_wino_wgrad_f32(stream(), B, H, H, Ci, Co, P(xc), P(dyc), P(dwf), None, 0, P(ws), ws.numel() * 4)
    assert_close(host(dw), host(dwf), tol, msg="%s bx wgrad vs f32 Winograd" % name)
