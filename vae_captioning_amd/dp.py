"""Data-parallel protocol (new work: the reference is single-GPU).

One process per GPU, replicated parameters, the minibatch sharded BY IMAGE (an image's nc
caption rows, its cluster-vector rows and its noise slices stay together).  Per step:

  1. the number of non-PAD labels of the GLOBAL batch: the CE loss divides by it (main.py:156-157) and the backward pass needs it
     as its scale.  With the global Q1 mix (below) it rides in that exchange's all-gather; otherwise (tower mix, --no_encoder) it
     is an all-reduce(sum) of ONE float.
  2. every rank runs forward + backward on its shard with the scales of `scales()`, so that
     the SUM over ranks of the local gradients is the gradient of the global-batch loss.
  3. ONE all-reduce(sum) of the flat gradient buffer (caption grads | tail scalars | VGG
     grads; with VGG fine-tuning issued as four asynchronous pieces, `gradient_buckets`).  The tail carries sum ||dX||^2 of the
     embedding IndexedSlices values (the un-deduplicated term of the global norm, quirk Q5) and the two REPORTING scalars ce_num
     and kl_sum: the reported losses are finalised behind the all-reduce (engine.apply_gradients), they have no collective of
     their own.
  4. identical clip + optimiser step on every replica (parameter-only terms such as the L2
     regulariser's gradient are applied once, inside the optimiser kernel).

Collectives per step: caption-only 3 with the global Q1 mix (all-gather, reduce-scatter, gradient all-reduce), 2 without; VGG16
fine-tuning 6 (all-gather, reduce-scatter, four gradient pieces).  STILL UNMEASURED on more than one GPU.

The one place the reference graph is not separable over rows is the Q1 reshape
(vae_model/decoder.py:109-110), which mixes the z samples of different batch rows.  Default
(q1_mode="global"): bit-for-bit the single-GPU semantics on the concatenated batch -- the per-row
[mean | std | label count] of every rank are all-gathered in ONE call (3 MB at 2560 rows), each rank samples the flat range
q in [rank*Nl*S, (rank+1)*Nl*S) of the global [S, Ng, L] tensor (q = s*Ng + n) that forms its own
z_rnn rows, and the [Ng, L] partial gradient sums [dmean | dstd] are reduce-scattered back to the owning ranks in ONE call.
q1_mode="tower" mixes inside each rank's shard instead (what N towers of the reference graph
would compute; equals the oracle with q1_groups = N) and needs no extra exchange.
"""
import numpy as np


def shard_batch(batch, rank, world, nc):
    """Rows of image i are i*nc .. i*nc+nc-1 (utils/caption_utils.py:4-25)."""
    some = batch["features"] if "features" in batch else batch["images"]
    B = some.shape[0]
    assert B % world == 0, "global batch must divide evenly over ranks"
    b0, b1 = rank * (B // world), (rank + 1) * (B // world)
    out = {}
    for k, v in batch.items():
        if k in ("features", "images"):
            out[k] = v[b0:b1]
        else:
            out[k] = v[b0 * nc:b1 * nc]
    return out


def shard_noise(noise, rank, world, n_rows_global, q1_mode="tower"):
    """Slices of the injected noise that belong to one rank.  eps [S, Ng, L]: "tower" takes the rank's
    rows of every sample; "global" takes the flat range q in [rank*Nl*S, (rank+1)*Nl*S) of q = s*Ng + n
    (the samples that land in this rank's z_rnn rows under the global Q1 reshape)."""
    nl = n_rows_global // world
    n0, n1 = rank * nl, (rank + 1) * nl
    out = {}
    for k, v in noise.items():
        if k == "c_means":
            out[k] = v
        elif k == "eps" and q1_mode == "global":
            S, Ng, L = v.shape
            out[k] = np.ascontiguousarray(v.reshape(S * Ng, L)[rank * nl * S:(rank + 1) * nl * S]).reshape(S, nl, L)
        elif k in ("eps", "drop_in", "drop_out"):
            out[k] = np.ascontiguousarray(v[:, n0:n1])
        else:
            out[k] = v[n0:n1]
    return out


def scales(n_local_rows, world, vector_loss):
    """(gscale, kl_scale_normal, kl_scale_ag, inv_n) for a shard of n_local_rows rows.
    gscale:  multiplies d(CE)/d(logits) (the AG vector loss differentiates N copies of rec_loss, Q3)
    kl_scale_*: multiplies ann * dKL in vc_latent_bwd_f32;  inv_n: 1 / global row count."""
    n_global = n_local_rows * world
    return (float(n_global) if vector_loss else 1.0, 0.1 / n_global, 0.1, 1.0 / n_global)


def gradient_buckets(n_cap, n_total, off_fc=None, off_c3=None):
    """[lo, hi) slices of the flat gradient buffer `gall` = [caption grads + tail scalars | VGG grads] in the order their
    all-reduces are issued.  Caption-only: the one buffer.  With VGG fine-tuning the ONE logical all-reduce is issued as four
    pieces in the order the gradients become final during the backward pass (so that RCCL runs under the convolution
    backward): caption side | fc1 + fc2 (89 % of the VGG bytes, final first) | conv3_1 .. conv5_3 | conv1_1 .. conv2_2.
    The slices are disjoint and cover [0, n_total) exactly (tests/test_dp_gloo.py)."""
    if off_fc is None or n_total == n_cap:
        return [(0, n_total)]
    return [(0, n_cap), (off_fc, n_total), (off_c3, off_fc), (n_cap, off_c3)]


class _Pending(object):
    """Handle of an asynchronous collective: wait() makes the CURRENT stream wait for it (no host block)."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        import torch
        torch.cuda.current_stream().wait_event(self.event)


class AbiComm(object):
    """The step's collectives through libvaecap's own RCCL entries (include/vaecap.h: vc_comm_*, vc_allreduce_sum_f32,
    vc_allgather_f32, vc_reducescatter_sum_f32) -- what a maintainer binding the C ABI gets; torch.distributed is not involved in
    the data path (it carries the 128-byte unique id to the other ranks: Trainer._agree_on_abi_comm).  Every collective of the communicator runs
    on ONE dedicated HIP stream in issue order; it waits for the issuing stream's work, and the issuing stream waits for it (blocking
    form) or for its event (`*_async(...).wait()`).  A failing call raises abi.VaecapError with RCCL's message: the step aborts."""
    _serial = 0

    def __init__(self, lib, world, rank, device, id_bytes):
        import ctypes
        import torch
        self.lib, self.world, self.rank = lib, int(world), int(rank)
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(id_bytes), 128)
        lib.vc_comm_init_rank(self.world, self.rank, buf, int(device), ctypes.byref(h))
        self.h = h
        self.stream = torch.cuda.Stream(device=device)
        ver = ctypes.c_int(0)
        lib.vc_comm_info(self.h, None, None, ctypes.byref(ver))
        self.rccl_version = int(ver.value)

    @staticmethod
    def unique_id(lib):
        import ctypes
        buf = ctypes.create_string_buffer(128)
        lib.vc_comm_unique_id(buf)
        return buf.raw

    @classmethod
    def single(cls, lib, device=0):
        """A one-rank communicator (tests; the collective branches then run through RCCL with nobody to talk to)."""
        return cls(lib, 1, 0, device, cls.unique_id(lib))

    def _issue(self, fn):
        import torch
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        self.stream.wait_event(ev)
        fn(self.stream.cuda_stream)
        done = torch.cuda.Event()
        done.record(self.stream)
        return _Pending(done)

    def all_reduce_async(self, t):
        assert t.is_contiguous() and t.dtype.is_floating_point and t.element_size() == 4
        return self._issue(lambda st: self.lib.vc_allreduce_sum_f32(self.h, st, t.data_ptr(), t.numel()))

    def all_reduce(self, t):
        self.all_reduce_async(t).wait()

    def all_gather(self, out, inp):
        assert out.numel() == inp.numel() * self.world and out.is_contiguous() and inp.is_contiguous()
        self._issue(lambda st: self.lib.vc_allgather_f32(self.h, st, inp.data_ptr(), out.data_ptr(), inp.numel())).wait()

    def reduce_scatter(self, out, inp):
        assert inp.numel() == out.numel() * self.world and out.is_contiguous() and inp.is_contiguous()
        self._issue(lambda st: self.lib.vc_reducescatter_sum_f32(self.h, st, inp.data_ptr(), out.data_ptr(), out.numel())).wait()

    def destroy(self):
        if self.h is not None:
            self.stream.synchronize()
            self.lib.vc_comm_destroy(self.h)

    def abort(self):
        self.lib.vc_comm_abort(self.h)
