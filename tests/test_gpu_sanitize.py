"""-m gpu: the C ABI's HOST code under UndefinedBehaviorSanitizer (there is no device-side sanitizer on this
stack; the kernels are guarded by the parity tests and by bit-reproducibility; AddressSanitizer cannot be used: ROCm's ASan runtime
intercepts the HSA allocator for its device mode and the HIP runtime then fails to allocate).  `make -C vae_captioning_amd/csrc
sanitize` (run by __graft_entry__.build) compiles every translation unit's host side -- argument checks, launch planners, image-range
loops, workspace sizing, the RCCL / roctx run-time binding, CRC-32C -- with -fsanitize=undefined (no recovery), libstdc++ assertions
and _FORTIFY_SOURCE; a child process with the UBSan runtime preloaded drives the hot path through that library: a fine-tune training step (every convolution planner), a
caption-only step with collectives on a one-rank communicator, error paths, and roctx ranges.  Any report fails the test."""
import glob
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_LIB = os.path.join(ROOT, "vae_captioning_amd", "lib", "libvaecap_san.so")

CHILD = r"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from vae_captioning_amd import abi, dp, spec, synth
from vae_captioning_amd.abi import VaecapError
from vae_captioning_amd.trainer import Trainer
from vae_captioning_amd.utils.parameters import Parameters
lib = abi.load(%(lib)r)
assert lib.vc_abi_version() == 4
os.environ["VC_TRACE"] = "1"
rng = np.random.default_rng(0)
# 1. fine-tune step: every convolution planner / image-range loop / workspace computation, three streams
p = Parameters(); p.fine_tune, p.batch_size, p.num_captions, p.gen_z_samples = True, 2, 2, 4
V = 300
tr = Trainer(p, V, lib=lib, seed=1)
tr.load_state_dict({**spec.init_caption_params(p, V, seed=1), **spec.init_vgg_params(seed=3)})
tr.set_batch(synth.make_batch(rng, 2, 2, 5, V, images=True, variable_len=True))
tr.train_step(); tr.train_step()
l1 = tr.losses(); assert all(np.isfinite(l1)), l1
del tr
# 2. caption-only AG step with the collective branches on a one-rank communicator (RCCL bound through dlopen), T changes between batches
p = Parameters(); p.prior, p.use_c_v = "AG", True
p.embed_size, p.encoder_hidden, p.decoder_hidden, p.latent_size, p.gen_z_samples, p.cnn_feature_size, p.num_captions = 32, 64, 64, 10, 4, 48, 3
tr = Trainer(p, 203, lib=lib, force_collectives=True)
tr.load_state_dict(spec.init_caption_params(p, 203, seed=2))
for T in (6, 9, 6):
    tr.set_batch(synth.make_batch(rng, 4, 3, T, 203, use_ci=True, variable_len=True, feature_size=48))
    tr.train_step()
assert all(np.isfinite(tr.losses()))
tr.comm.destroy()
try:
    tr.train_step(); raise SystemExit("a destroyed communicator was accepted")
except VaecapError:
    pass
# 3. error paths and host utilities
for bad in (lambda: lib.vc_gemm_f32(None, 0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, None, 0),
            lambda: lib.vc_conv3x3_wino_fwd_f32(None, 2, 7, 8, 32, 64, 1, 1, None, 1, None, 0),
            lambda: lib.vc_allreduce_sum_f32(ctypes.create_string_buffer(64), None, None, 4)):
    try:
        bad(); raise SystemExit("an invalid call was accepted")
    except VaecapError:
        pass
crc = ctypes.c_uint32(0)
lib.vc_host_crc32c(b"123456789", 9, ctypes.byref(crc)); assert crc.value == 0xE3069283
for shape in ((64, 224, 224, 64, 64), (512, 224, 224, 64, 64), (5, 14, 14, 96, 128), (3, 6, 10, 48, 32)):
    lib.vc_conv3x3_wino_supported(*shape, 0); lib.vc_conv3x3_wino_wgrad_workspace_bytes(*shape); lib.vc_conv3x3_wino_mask_words(*shape[:3], shape[4])
    lib.vc_gemm_workspace_bytes(shape[0], shape[3] * 64, shape[4] * 9)
torch.cuda.synchronize()
print("SANITIZED RUN OK")
"""


def test_host_code_of_the_abi_under_ubsan(tmp_path):
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so"))
    if not os.path.exists(SAN_LIB) or not rt:
        pytest.skip("sanitizer build (make -C vae_captioning_amd/csrc sanitize) or the UBSan runtime is not present")
    script = tmp_path / "child.py"
    script.write_text(CHILD % dict(root=ROOT, lib=SAN_LIB))
    env = dict(os.environ, LD_PRELOAD=rt[-1], UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=1200, cwd=tmp_path)
    out = r.stdout + r.stderr
    assert "runtime error:" not in out and "Sanitizer" not in out, out[-4000:]
    assert r.returncode == 0 and "SANITIZED RUN OK" in r.stdout, out[-4000:]
