"""-m gpu: the C ABI's multi-GPU entries (include/vaecap.h: vc_comm_*, vc_allreduce_sum_f32, vc_allgather_f32, vc_reducescatter_sum_f32;
csrc/comm.hip over RCCL) on the one GPU a test box has: a ONE-RANK communicator (sum over one rank = identity; the calls still go
through RCCL's launch path on the communicator's stream), the Trainer's collective branches through it, and the failure paths (a
destroyed / foreign handle, a duplicate GPU) -- every one a non-zero return code with a message, never a crash, and the step aborts.
RCCL with more than one rank needs more than one GPU: UNMEASURED here; the protocol above the collectives is
covered by tests/test_dp_gloo.py (world 2, CPU) and the emulated ranks of tests/test_gpu_fullsize.py."""
import ctypes

import numpy as np
import pytest
import torch

from vae_captioning_amd import abi, dp, spec, synth
from vae_captioning_amd.abi import VaecapError
from vae_captioning_amd.trainer import Trainer
from vae_captioning_amd.utils.parameters import Parameters

from .gpu_util import grads_only

pytestmark = pytest.mark.gpu


def test_collectives_on_a_one_rank_communicator(lib):
    assert lib.vc_comm_available() == 1
    comm = dp.AbiComm.single(lib)
    assert comm.rccl_version > 20000
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(1 << 20, device="cuda", generator=g)
    ref = x.clone()
    comm.all_reduce(x)                                   # blocking form: the current stream waits for the communicator's stream
    assert torch.equal(x, ref)
    h = comm.all_reduce_async(x)                         # asynchronous form (the gradient buckets)
    h.wait()
    assert torch.equal(x, ref)
    out = torch.zeros(4096, device="cuda")
    comm.all_gather(out, ref[:4096])
    assert torch.equal(out, ref[:4096])
    out.zero_()
    comm.reduce_scatter(out, ref[4096:8192])
    assert torch.equal(out, ref[4096:8192])
    lib.vc_allreduce_sum_f32(comm.h, torch.cuda.current_stream().cuda_stream, None, 0)   # empty: accepted
    comm.destroy()
    comm.destroy()                                       # idempotent


def test_failures_are_return_codes_not_crashes(lib):
    comm = dp.AbiComm.single(lib)
    x = torch.ones(16, device="cuda")
    comm.destroy()
    with pytest.raises(VaecapError, match="destroyed or aborted"):
        comm.all_reduce(x)
    with pytest.raises(VaecapError, match="destroyed or aborted"):
        comm.all_gather(torch.empty(16, device="cuda"), x)
    bogus = ctypes.create_string_buffer(64)              # a pointer that is not a communicator handle
    with pytest.raises(VaecapError, match="not a communicator handle"):
        lib.vc_allreduce_sum_f32(bogus, torch.cuda.current_stream().cuda_stream, x.data_ptr(), 16)
    with pytest.raises(VaecapError, match="not a communicator handle"):
        lib.vc_reducescatter_sum_f32(None, None, x.data_ptr(), x.data_ptr(), 1)
    h = ctypes.c_void_p()
    with pytest.raises(VaecapError, match="bad argument"):
        lib.vc_comm_init_rank(2, 5, ctypes.create_string_buffer(128), 0, ctypes.byref(h))   # rank outside the world
    assert torch.equal(x, torch.ones(16, device="cuda"))


def _small():
    p = Parameters()
    p.embed_size, p.encoder_hidden, p.decoder_hidden = 32, 64, 96
    p.latent_size, p.gen_z_samples, p.cnn_feature_size = 12, 5, 40
    p.num_captions, p.batch_size = 3, 4
    return p


def test_trainer_collective_branches_through_the_abi_communicator(lib):
    """force_collectives on one rank with NO torch.distributed process group: the count / loss-scalar / gradient all-reduces and the
    Q1 all-gather / reduce-scatter run through vc_* (Trainer(comm='auto') -> dp.AbiComm.single); parameters after two steps equal
    the collective-free trainer bit for bit (sum over one rank)."""
    assert not torch.distributed.is_initialized()
    p = _small()
    V, B, T = 203, 4, 6
    rng = np.random.default_rng(13)
    P0 = spec.init_caption_params(p, V, seed=14)
    batch = synth.make_batch(rng, B, p.num_captions, T, V, variable_len=True, feature_size=p.cnn_feature_size)
    noise = synth.make_noise(rng, B * p.num_captions, T, p)
    res = []
    for force in (False, True):
        tr = Trainer(p, V, lib=lib, force_collectives=force)
        assert (tr.comm is not None) == force
        tr.cap.q1_mode = "global"
        tr.load_state_dict(P0)
        for _ in range(2):
            tr.set_batch(batch, noise)
            tr.train_step()
        res.append((tr.losses(), tr.state_dict()))
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        np.testing.assert_array_equal(res[0][1][k], res[1][1][k], err_msg=k)


def test_fine_tune_gradient_buckets_through_the_abi_communicator(lib):
    """The four asynchronous all-reduce pieces of the fine-tune step (caption | fc | conv3..5 | conv1..2) on the communicator's stream under
    the three-stream convolution backward: bit-identical to the single blocking all-reduce and to the collective-free step."""
    p = Parameters()
    p.fine_tune, p.batch_size, p.num_captions, p.gen_z_samples = True, 2, 2, 4
    V = 300
    rng = np.random.default_rng(4)
    batch = synth.make_batch(rng, 2, 2, 5, V, images=True, variable_len=True)
    P0 = {**spec.init_caption_params(p, V, seed=1), **spec.init_vgg_params(seed=3)}
    res = []
    for force, buckets in ((False, False), (True, False), (True, True)):
        tr = Trainer(p, V, lib=lib, force_collectives=force, seed=5)
        tr.buckets = buckets
        tr.load_state_dict(P0)
        tr.set_batch(batch)
        tr.train_step()
        res.append((tr.losses(), grads_only(tr)))
        del tr
    assert res[0][0] == res[1][0] == res[2][0]
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][1], res[2][1])


def test_a_dead_communicator_aborts_the_step(lib):
    """destroyed communicator -> non-zero return code -> VaecapError out of train_step; no parameter is updated by the aborted step."""
    p = _small()
    V, B, T = 203, 4, 6
    rng = np.random.default_rng(15)
    P0 = spec.init_caption_params(p, V, seed=16)
    batch = synth.make_batch(rng, B, p.num_captions, T, V, variable_len=True, feature_size=p.cnn_feature_size)
    tr = Trainer(p, V, lib=lib, force_collectives=True)
    tr.load_state_dict(P0)
    tr.set_batch(batch)
    tr.train_step()
    after_one = tr.state_dict()
    tr.comm.destroy()
    with pytest.raises(VaecapError, match="vc_allreduce_sum_f32 failed .*destroyed or aborted"):
        tr.train_step()
    torch.cuda.synchronize()
    now = tr.state_dict()
    for k in after_one:
        np.testing.assert_array_equal(after_one[k], now[k], err_msg=k)


def test_losses_of_a_data_parallel_training_forward_raise_until_the_gradients_are_applied(lib):
    """With collectives on, the reported kld / rec_loss / lower_bound of a TRAINING forward ride in the tail of the gradient all-reduce:
    between forward(train=True) and apply_gradients() `losses()` raises instead of returning the previous step's values; an evaluation
    forward (train=False: scalars reduced on the spot, the reference's validate(), main.py:262-284) clears the state, and so does the
    completed step."""
    p = _small()
    V, B, T = 203, 4, 6
    rng = np.random.default_rng(17)
    batch = synth.make_batch(rng, B, p.num_captions, T, V, variable_len=True, feature_size=p.cnn_feature_size)
    noise = synth.make_noise(rng, B * p.num_captions, T, p)
    tr = Trainer(p, V, lib=lib, force_collectives=True)
    tr.load_state_dict(spec.init_caption_params(p, V, seed=18))
    tr.set_batch(batch, noise)
    tr.train_step()
    done = tr.losses()                      # a completed step reports
    tr.cap.forward(train=True)
    with pytest.raises(RuntimeError, match="final only after apply_gradients"):
        tr.cap.losses()
    tr.cap.forward(train=False)             # abandoning the step for an evaluation pass is legal and readable
    ev = tr.cap.losses()
    assert np.isfinite(ev[1]) and ev[1] != done[1]
    tr.train_step()                         # and a whole step afterwards reports again
    assert np.isfinite(tr.losses()[1])


def test_set_batch_under_a_captured_graph_names_shapes_and_dtypes(lib):
    """a hipGraph bakes its input buffers: a later batch of another shape OR dtype is refused with both in the message (uint8 images
    against float32 ones have identical shapes)"""
    p = _small()
    V, B, T = 203, 4, 6
    rng = np.random.default_rng(19)
    batch = synth.make_batch(rng, B, p.num_captions, T, V, feature_size=p.cnn_feature_size)
    tr = Trainer(p, V, lib=lib)
    tr.load_state_dict(spec.init_caption_params(p, V, seed=20))
    tr.set_batch(batch)
    tr.capture()
    tr.train_step()
    longer = synth.make_batch(rng, B, p.num_captions, T + 2, V, feature_size=p.cnn_feature_size)
    with pytest.raises(ValueError, match=r"changed from \(.*\) \w+ to \(.*\) \w+"):
        tr.set_batch(longer)
