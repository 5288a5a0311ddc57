"""Oracle: VGG16 feature extractor (utils/image_embeddings.py:26-238) forward and
backward in numpy.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Layout NHWC activations, HWIO kernels (image_embeddings.py:38-40).
TF-sem.: conv2d stride 1 padding SAME with a 3x3 kernel = 1 zero pixel each side;
max_pool 2x2/2 SAME on even dims = no padding; MaxPoolGrad routes the gradient to
the first maximum in (dy, dx) scan order (ties only matter at exact equality;
after ReLU the tied-at-zero case gets zero gradient from ReluGrad anyway).
"""
import numpy as np

MEAN_RGB = np.array([123.68, 116.779, 103.939], np.float32)  # image_embeddings.py:31-34

# (name, Cin, Cout) in order; 'P' = maxpool.  image_embeddings.py:36-212
LAYERS = [("conv1_1", 3, 64), ("conv1_2", 64, 64), "P",
          ("conv2_1", 64, 128), ("conv2_2", 128, 128), "P",
          ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), "P",
          ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), "P",
          ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512), "P"]


def var_names(layer):
    """Quirk Q17: conv5_x variables are weights_conv / biases_conv."""
    if layer.startswith("conv5"):
        return "cnn/%s/weights_conv" % layer, "cnn/%s/biases_conv" % layer
    return "cnn/%s/weights" % layer, "cnn/%s/biases" % layer


def _shift(x, dy, dx):
    """y[b, h, w] = x[b, h+dy, w+dx] with zero fill."""
    B, H, W, C = x.shape
    out = np.zeros_like(x)
    hs, he = max(0, -dy), min(H, H - dy)
    ws, we = max(0, -dx), min(W, W - dx)
    out[:, hs:he, ws:we] = x[:, hs + dy:he + dy, ws + dx:we + dx]
    return out


def conv3x3_fwd(x, w, b):
    B, H, W, Ci = x.shape
    Co = w.shape[3]
    out = np.zeros((B * H * W, Co), x.dtype)
    for ky in range(3):
        for kx in range(3):
            out += _shift(x, ky - 1, kx - 1).reshape(-1, Ci) @ w[ky, kx]
    return out.reshape(B, H, W, Co) + b


def conv3x3_bwd(x, w, dy, need_dx=True):
    B, H, W, Ci = x.shape
    Co = w.shape[3]
    dyf = dy.reshape(-1, Co)
    dw = np.zeros_like(w)
    dx = np.zeros_like(x) if need_dx else None
    for ky in range(3):
        for kx in range(3):
            dw[ky, kx] = _shift(x, ky - 1, kx - 1).reshape(-1, Ci).T @ dyf
            if need_dx:
                dx += _shift((dyf @ w[ky, kx].T).reshape(B, H, W, Ci), 1 - ky, 1 - kx)
    return dx, dw, dyf.sum(axis=0)


def maxpool_fwd(x):
    B, H, W, C = x.shape
    xr = x.reshape(B, H // 2, 2, W // 2, 2, C).transpose(0, 1, 3, 2, 4, 5).reshape(B, H // 2, W // 2, 4, C)
    arg = xr.argmax(axis=3)  # first max in (dy, dx) scan order
    return np.take_along_axis(xr, arg[:, :, :, None, :], axis=3)[:, :, :, 0, :], arg


def maxpool_bwd(dy, arg, in_shape):
    B, H, W, C = in_shape
    d = np.zeros((B, H // 2, W // 2, 4, C), dy.dtype)
    np.put_along_axis(d, arg[:, :, :, None, :], dy[:, :, :, None, :], axis=3)
    return d.reshape(B, H // 2, W // 2, 2, 2, C).transpose(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)


def forward(P, images, drop1=None, drop2=None, keep=1.0):
    """images [B, 224, 224, 3] float32 RGB (0..255).  Returns fc2 and a cache."""
    x = images - MEAN_RGB.astype(images.dtype)
    cache = []
    for l in LAYERS:
        if l == "P":
            y, arg = maxpool_fwd(x)
            cache.append(("P", x.shape, arg))
            x = y
        else:
            wn, bn = var_names(l[0])
            y = np.maximum(conv3x3_fwd(x, P[wn], P[bn]), 0)
            cache.append((l[0], x, y))
            x = y
    B = x.shape[0]
    flat = x.reshape(B, -1)  # NHWC flatten order, image_embeddings.py:222
    fc1 = np.maximum(flat @ P["cnn/fc1/weights"] + P["cnn/fc1/biases"], 0)
    fc1d = fc1 * drop1 / images.dtype.type(keep) if drop1 is not None else fc1
    fc2 = np.maximum(fc1d @ P["cnn/fc2/weights"] + P["cnn/fc2/biases"], 0)
    fc2d = fc2 * drop2 / images.dtype.type(keep) if drop2 is not None else fc2
    return fc2d, dict(conv=cache, pool5_shape=x.shape, flat=flat, fc1=fc1, fc1d=fc1d, fc2=fc2,
                      drop1=drop1, drop2=drop2, keep=keep)


def backward(P, cache, dfc2):
    """Gradient of the non-regulariser loss w.r.t. every cnn/* variable."""
    G = {}
    keep = cache["keep"]
    d = dfc2
    if cache["drop2"] is not None:
        d = d * cache["drop2"] / d.dtype.type(keep)
    d = d * (cache["fc2"] > 0)
    G["cnn/fc2/weights"] = cache["fc1d"].T @ d
    G["cnn/fc2/biases"] = d.sum(axis=0)
    d = d @ P["cnn/fc2/weights"].T
    if cache["drop1"] is not None:
        d = d * cache["drop1"] / d.dtype.type(keep)
    d = d * (cache["fc1"] > 0)
    G["cnn/fc1/weights"] = cache["flat"].T @ d
    G["cnn/fc1/biases"] = d.sum(axis=0)
    d = (d @ P["cnn/fc1/weights"].T).reshape(cache["pool5_shape"])
    conv = cache["conv"]
    for li in range(len(conv) - 1, -1, -1):
        ent = conv[li]
        if ent[0] == "P":
            d = maxpool_bwd(d, ent[2], ent[1])
        else:
            name, x, y = ent
            wn, bn = var_names(name)
            d = d * (y > 0)
            d, G[wn], G[bn] = conv3x3_bwd(x, P[wn], d, need_dx=(li > 0))
    return G


def l2_reg_loss(P, weight_decay):
    """main.py:69-74 (Q9): l2_regularizer(wd) on every cnn/* variable, biases
    included: wd * sum(w^2)/2 each."""
    tot = 0.0
    for n, w in P.items():
        if n.startswith("cnn/"):
            tot += float(np.sum(w.astype(np.float64) ** 2)) / 2
    return np.float32(weight_decay * tot)


def load_order():
    """Quirk Q18 (image_embeddings.py:240-246): the first 30 alphabetically sorted
    npz keys map onto self.parameters in construction order."""
    order = []
    for l in LAYERS:
        if l != "P":
            order += list(var_names(l[0]))
    order += ["cnn/fc1/weights", "cnn/fc1/biases", "cnn/fc2/weights", "cnn/fc2/biases"]
    return order
