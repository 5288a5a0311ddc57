"""Host-side image loading (vae_captioning_amd/utils/image_utils.py; utils/image_utils.py:5-13,
gen_caption.py:57-71).  cv2 / Keras are absent: property tests of the restated resize."""
import numpy as np

from vae_captioning_amd.utils.image_utils import keras_load_img, load_image, resize_bilinear_u8


def test_bilinear_resize_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    assert np.array_equal(resize_bilinear_u8(img, (53, 37)), img)                 # same size: identity
    big = rng.integers(0, 256, size=(448, 448, 3), dtype=np.uint8)
    half = resize_bilinear_u8(big, (224, 224))                                     # exact 1/2: 2x2 box mean, rounded
    box = (big.reshape(224, 2, 224, 2, 3).astype(np.int64).sum(axis=(1, 3)) + 2) >> 2
    assert half.shape == (224, 224, 3) and np.abs(half.astype(np.int64) - box).max() <= 1
    const = np.full((100, 80, 3), 77, np.uint8)
    assert np.all(resize_bilinear_u8(const, (224, 224)) == 77)
    up = resize_bilinear_u8(img, (224, 224))
    assert up.dtype == np.uint8 and up.min() >= img.min() and up.max() <= img.max()
    assert resize_bilinear_u8(img[:, :, 0], (10, 20)).shape == (20, 10, 1)         # (width, height) argument order


def test_load_image_and_keras_load_img(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, size=(60, 90, 3), dtype=np.uint8)
    Image.fromarray(a).save(tmp_path / "a.png")
    Image.fromarray(a[:, :, 0]).save(tmp_path / "g.png")                            # grayscale file
    out = load_image(str(tmp_path / "a.png"))
    assert out.shape == (224, 224, 3) and out.dtype == np.uint8
    assert np.array_equal(out, resize_bilinear_u8(a, (224, 224)))                   # RGB order kept
    g = load_image(str(tmp_path / "g.png"))
    assert np.array_equal(g[..., 0], g[..., 1]) and np.array_equal(g[..., 1], g[..., 2])
    x, im = keras_load_img(str(tmp_path / "a.png"))
    assert x.shape == (1, 224, 224, 3) and x.dtype == np.float32
    # NEAREST: every output pixel is one of the source pixels of its row/column neighbourhood
    assert set(np.unique(x)).issubset(set(np.unique(a).astype(np.float32)))
    same, _ = keras_load_img(str(tmp_path / "a.png"), target_size=(60, 90))
    assert np.array_equal(same[0], a.astype(np.float32))


def test_bilinear_resize_geometry_matches_torch_interpolate():
    """cv2.resize(INTER_LINEAR) cannot run here; its sampling GEOMETRY (half-pixel centres, edge clamping, no anti-aliasing) is what
    torch.nn.functional.interpolate(mode='bilinear', align_corners=False, antialias=False) implements in floating point.  The 8-bit
    restatement (11-bit fixed-point coefficients, two-pass rounding) must agree with it to within ONE grey level for up- and
    down-scaling, odd sizes and non-square targets -- a wrong centre convention or tap index shows up as errors of tens of levels."""
    import torch
    from vae_captioning_amd.utils.image_utils import resize_bilinear_u8
    rng = np.random.default_rng(3)
    for (sh, sw), (dw, dh) in (((37, 53), (224, 224)), ((480, 640), (224, 224)), ((300, 200), (224, 224)), ((224, 224), (97, 131)), ((5, 7), (16, 9))):
        img = rng.integers(0, 256, size=(sh, sw, 3), dtype=np.uint8)
        got = resize_bilinear_u8(img, (dw, dh)).astype(np.int64)
        t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
        ref = torch.nn.functional.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        assert got.shape == (dh, dw, 3)
        assert np.abs(got - ref).max() <= 1.0, ((sh, sw), (dw, dh), np.abs(got - ref).max())
