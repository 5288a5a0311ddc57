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
