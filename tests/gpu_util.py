"""Helpers for the -m gpu parity tests: everything goes through the C ABI."""
import numpy as np
import torch

from vae_captioning_amd import abi


def stream():
    return torch.cuda.current_stream().cuda_stream


# Tensors created by dev() stay alive until the end of the test: `P(dev(x))` hands a raw
# pointer to the library, and a temporary freed before the next dev() call would have its
# block recycled by the caching allocator (the next upload would overwrite it).
_KEEP = []


def dev(x, dtype=None):
    a = np.ascontiguousarray(x)
    if dtype is not None:
        a = a.astype(dtype)
    t = torch.from_numpy(a).cuda()
    _KEEP.append(t)
    return t


def release():
    del _KEEP[:]


def zeros(*shape, dtype=torch.float32):
    return torch.zeros(*shape, dtype=dtype, device="cuda")


def empty_bytes(nbytes):
    return torch.empty(max(int(nbytes), 16) // 4 + 4, dtype=torch.float32, device="cuda")


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


P = abi.ptr


def assert_close(got, ref, rtol, atol_scale=None, msg=""):
    """max|got-ref| <= rtol * max|ref| (+ tiny); prints where the worst element is."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (msg, got.shape, ref.shape)
    scale = np.abs(ref).max() if atol_scale is None else atol_scale
    err = np.abs(got - ref)
    if err.size == 0:
        return
    worst = np.unravel_index(np.argmax(err), err.shape)
    tol = rtol * scale + 1e-30
    assert np.isfinite(got).all(), "%s: non-finite output" % msg
    assert err.max() <= tol, "%s: max err %.3e > tol %.3e at %s (got %r ref %r), %d/%d elements off" % (
        msg, err.max(), tol, worst, got[worst], ref[worst], int((err > tol).sum()), err.size)


# ---- the C4 activation layout of the Winograd kernels (include/vaecap.h): [B][C/4][H][W][4] ----
def to_c4(a):
    """NHWC numpy array [B,H,W,C] -> C4 [B, C/4, H, W, 4]."""
    a = np.asarray(a)
    B, H, W, C = a.shape
    return np.ascontiguousarray(a.reshape(B, H, W, C // 4, 4).transpose(0, 3, 1, 2, 4))


def from_c4(a, shape=None):
    """C4 array (any shape holding B*C/4*H*W*4 elements; `shape` = the NHWC shape [B,H,W,C]) -> NHWC numpy array."""
    a = np.asarray(a)
    if shape is None:
        B, C4, H, W, _ = a.shape
        shape = (B, H, W, C4 * 4)
    B, H, W, C = shape
    return np.ascontiguousarray(a.reshape(B, C // 4, H, W, 4).transpose(0, 2, 3, 1, 4).reshape(B, H, W, C))


def dev_c4(x, dtype=None):
    return dev(to_c4(x), dtype)


def host_c4(t, shape):
    """device tensor holding a C4 activation -> NHWC numpy array of `shape` [B,H,W,C]."""
    return from_c4(host(t), shape)


def grads_only(tr):
    """Trainer.gall with the two REPORTING scalars of the data-parallel step (ce_num, kl_sum: tail slots 1 and 2, written only when
    collectives are on) zeroed -- what must be identical between a collective and a collective-free step."""
    g = tr.gall.clone()
    n = tr.cap.store.n
    g[n + 1:n + 3] = 0
    return g
