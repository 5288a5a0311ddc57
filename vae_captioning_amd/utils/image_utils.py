"""Image loading for the two places the reference decodes image files (host side, PIL + numpy;
OpenCV and Keras are not in this image):

  load_image       utils/image_utils.py:5-13 (cv2.imread -> cv2.resize(img, shape) -> BGR2RGB), used by
                   preprocess.py:27-28 to fill the HDF5 image array.
  keras_load_img   gen_caption.py:67-69 (keras load_img(target_size=(224, 224)) + img_to_array): PIL decode,
                   RGB, NEAREST-neighbour resize (Image.resize's default in the Keras/PIL versions of the time).

`resize_bilinear_u8` restates OpenCV's 8-bit INTER_LINEAR resize (half-pixel centres, 11-bit fixed-point
coefficients, its two-pass rounding) from the published algorithm; it cannot be checked against cv2 here
-- the tests pin the properties that do not need cv2 (identity at equal size, exact 2x2 box means at integer
1/2 scale, constant images, range) and, since round 3, the sampling geometry against a third-party
implementation of the same convention (torch.nn.functional.interpolate, bilinear, half-pixel centres, no
anti-aliasing: within one grey level for up- and down-scaling).  The exact fixed-point rounding stays
UNVERIFIED against cv2."""
import numpy as np


def _pil():
    from PIL import Image
    return Image


def resize_bilinear_u8(img, shape):
    """img [H, W, C] uint8 -> [shape[1], shape[0], C] uint8 (cv2.resize(img, (width, height)) argument order)."""
    img = np.asarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    sh, sw = img.shape[:2]
    dw, dh = int(shape[0]), int(shape[1])

    def taps(dn, sn):
        scale = sn / float(dn)
        f = (np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5
        i0 = np.floor(f).astype(np.int64)
        f = (f - i0).astype(np.float32)
        lo = i0 < 0
        f[lo], i0[lo] = 0.0, 0
        hi = i0 >= sn - 1
        f[hi], i0[hi] = 0.0, sn - 1
        i1 = np.minimum(i0 + 1, sn - 1)
        c1 = np.rint(f * 2048.0).astype(np.int64)          # saturate_cast<short>(fx * INTER_RESIZE_COEF_SCALE)
        c0 = np.rint((1.0 - f) * 2048.0).astype(np.int64)
        return i0, i1, c0, c1

    x0, x1, a0, a1 = taps(dw, sw)
    y0, y1, b0, b1 = taps(dh, sh)
    src = img.astype(np.int64)
    rows = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]     # horizontal pass, scale 2^11
    s0, s1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def load_image(image_path, shape=(224, 224)):
    """-> [shape[1], shape[0], 3] uint8 RGB (grayscale files become three equal channels, as cv2.imread's
    default IMREAD_COLOR does)."""
    im = _pil().open(image_path).convert("RGB")
    return resize_bilinear_u8(np.asarray(im), shape)


def keras_load_img(image_path, target_size=(224, 224)):
    """-> (x [1, H, W, 3] float32 RGB 0..255, PIL image): load_img + img_to_array + expand_dims of
    gen_caption.py:67-70.  (preprocess_input's RGB->BGR swap and mean subtraction belong to the Keras
    VGG16 weights; this build's VGG16 takes RGB and subtracts the RGB means on device.)"""
    Image = _pil()
    im = Image.open(image_path).convert("RGB")
    if target_size is not None and im.size != (target_size[1], target_size[0]):
        im = im.resize((target_size[1], target_size[0]), Image.NEAREST)
    return np.asarray(im, dtype=np.float32)[None], im
