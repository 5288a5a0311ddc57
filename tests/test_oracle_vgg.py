"""Oracle self-checks (CPU) for the VGG16 restatement against torch-CPU
conv2d / max_pool2d / autograd."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import vgg
from vae_captioning_amd import spec


def test_conv3x3_fwd_bwd_vs_torch():
    rng = np.random.default_rng(0)
    x = rng.normal(size=(2, 6, 5, 3))
    w = rng.normal(size=(3, 3, 3, 4))
    b = rng.normal(size=(4,))
    dy = rng.normal(size=(2, 6, 5, 4))
    y = vgg.conv3x3_fwd(x, w, b)
    dx, dw, db = vgg.conv3x3_bwd(x, w, dy)
    tx = torch.tensor(x.transpose(0, 3, 1, 2), requires_grad=True)
    tw = torch.tensor(w.transpose(3, 2, 0, 1), requires_grad=True)
    tb = torch.tensor(b, requires_grad=True)
    ty = F.conv2d(tx, tw, tb, padding=1)
    ty.backward(torch.tensor(dy.transpose(0, 3, 1, 2)))
    np.testing.assert_allclose(y, ty.detach().numpy().transpose(0, 2, 3, 1), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(dx, tx.grad.numpy().transpose(0, 2, 3, 1), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(dw, tw.grad.numpy().transpose(2, 3, 1, 0), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(db, tb.grad.numpy(), rtol=1e-10)


def test_maxpool_fwd_bwd_vs_torch():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(2, 6, 4, 3))
    dy = rng.normal(size=(2, 3, 2, 3))
    y, arg = vgg.maxpool_fwd(x)
    dx = vgg.maxpool_bwd(dy, arg, x.shape)
    tx = torch.tensor(x.transpose(0, 3, 1, 2), requires_grad=True)
    ty = F.max_pool2d(tx, 2, 2)
    ty.backward(torch.tensor(dy.transpose(0, 3, 1, 2)))
    np.testing.assert_array_equal(y, ty.detach().numpy().transpose(0, 2, 3, 1))
    np.testing.assert_array_equal(dx, tx.grad.numpy().transpose(0, 2, 3, 1))


def test_load_order_matches_sorted_npz_keys():
    """Q18: sorted npz keys conv1_1_W, conv1_1_b, ..., fc6_W, fc6_b, fc7_W, fc7_b, fc8_*"""
    keys = []
    for l in vgg.LAYERS:
        if l != "P":
            keys += [l[0] + "_W", l[0] + "_b"]
    keys += ["fc6_W", "fc6_b", "fc7_W", "fc7_b", "fc8_W", "fc8_b"]
    assert sorted(keys) == keys
    order = vgg.load_order()
    assert len(order) == 30 and order[0] == "cnn/conv1_1/weights" and order[25] == "cnn/conv5_3/biases_conv"
    assert order == [n for n, _ in spec.vgg_variables()]


def test_full_stack_forward_backward_vs_torch_fp64():
    """B=1, 224x224: whole VGG16 fwd+bwd (with injected fc dropout masks).  fp64:
    in fp32 a handful of ReLU / argmax decisions flip between two summation
    orders, which moves single gradient entries by ~1 % of the tensor max."""
    P = {k: v.astype(np.float64) for k, v in spec.init_vgg_params(seed=3).items()}
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(1, 224, 224, 3)).astype(np.float64)
    d1 = (rng.random((1, 4096)) < 0.5).astype(np.float64)
    d2 = (rng.random((1, 4096)) < 0.5).astype(np.float64)
    dfc2 = rng.normal(size=(1, 4096))
    fc2, cache = vgg.forward(P, img, d1, d2, keep=0.5)
    G = vgg.backward(P, cache, dfc2)

    tP = {k: torch.tensor(v, requires_grad=True) for k, v in P.items()}
    x = torch.tensor(img - vgg.MEAN_RGB.astype(np.float64)).permute(0, 3, 1, 2)
    for l in vgg.LAYERS:
        if l == "P":
            x = F.max_pool2d(x, 2, 2)
        else:
            wn, bn = vgg.var_names(l[0])
            x = F.relu(F.conv2d(x, tP[wn].permute(3, 2, 0, 1), tP[bn], padding=1))
    flat = x.permute(0, 2, 3, 1).reshape(1, -1)
    f1 = F.relu(flat @ tP["cnn/fc1/weights"] + tP["cnn/fc1/biases"]) * torch.tensor(d1) / 0.5
    f2 = F.relu(f1 @ tP["cnn/fc2/weights"] + tP["cnn/fc2/biases"]) * torch.tensor(d2) / 0.5
    f2.backward(torch.tensor(dfc2))
    np.testing.assert_allclose(fc2, f2.detach().numpy(), rtol=1e-9, atol=1e-9 * np.abs(fc2).max())
    for n in G:
        ref = tP[n].grad.numpy()
        tol = 1e-10 * np.abs(ref).max() + 1e-300
        assert np.abs(G[n] - ref).max() <= tol, (n, np.abs(G[n] - ref).max(), tol)
