"""The workgroup orders and the stream placement chosen in round 4 change WHEN a piece of work runs and on which XCD, never what it
computes: every A/B switch (read once per process, hence the subprocesses) must reproduce the default's results bit for bit.
  VC_WINO4_TG     chunk width of the F(4x4,3x3) workgroup order (csrc/conv_wino4.hip wino4_launch; 0 = all channel tiles together)
  VC_WGRAD_XCD    XCD-contiguous (split, tile) ranges of the weight gradient (csrc/conv_wino_wgrad_kernel.h)
  VC_LOGITS_DW    where the logits layer's kernel gradient is issued (engine.backward)
  VC_WINO4V       conv4_x / conv5_x on the once-transformed input (csrc/conv_wino4.hip MODE 2) or on the fused kernel (0)
"""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KERNELS = r"""
import hashlib, json, sys, torch
sys.path.insert(0, %r)
from vae_captioning_amd import abi
from vae_captioning_amd.abi import ptr as P
lib = abi.load()
st = lambda: torch.cuda.current_stream().cuda_stream
out = {}
torch.manual_seed(7)
for (B, H, ci, co) in ((3, 28, 32, 256), (2, 14, 64, 512), (2, 56, 128, 128)):
    x = torch.randn(B * H * H * ci, device="cuda")
    w = torch.randn(3, 3, ci, co, device="cuda") * 0.05
    bias = torch.randn(co, device="cuda")
    y = torch.empty(B * H * H * co, device="cuda")
    mk = torch.zeros(lib.vc_conv3x3_wino4_mask_words(B, H, H, co), dtype=torch.int32, device="cuda")
    vp = torch.empty(36 * ci * co, device="cuda")
    assert lib.vc_conv3x3_wino4_pack_f32(st(), ci, co, P(w), 0, P(vp)) == 0
    assert lib.vc_conv3x3_wino4_fwd_mask_f32(st(), B, H, H, ci, co, P(x), P(vp), P(bias), P(y), 1, P(mk)) == 0
    # the NEXT layer's data gradient (co -> c2 channels) reads the bits the forward left (indexed by workgroup: both kernels walk the same order)
    c2 = 64
    w2 = torch.randn(3, 3, co, c2, device="cuda") * 0.05
    dy2 = torch.randn(B * H * H * c2, device="cuda")
    dx = torch.empty(B * H * H * co, device="cuda")
    vpt = torch.empty(36 * co * c2, device="cuda")
    assert lib.vc_conv3x3_wino4_pack_f32(st(), co, c2, P(w2), 1, P(vpt)) == 0
    assert lib.vc_conv3x3_wino4_dgrad_bits_f32(st(), B, H, H, co, c2, P(dy2), P(vpt), P(mk), P(dx)) == 0
    torch.cuda.synchronize()
    key = "%%dx%%dx%%d->%%d" %% (B, H, ci, co)
    out["fwd " + key] = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()
    out["dgrad " + key] = hashlib.sha256(dx.cpu().numpy().tobytes()).hexdigest()
for (B, H, ci, co) in ((2, 28, 128, 256), (1, 14, 512, 512), (3, 56, 64, 128)):
    x = torch.randn(B * H * H * ci, device="cuda")
    dy = torch.randn(B * H * H * co, device="cuda")
    dw, db = torch.empty(9 * ci * co, device="cuda"), torch.empty(co, device="cuda")
    nb = lib.vc_conv3x3_wino_wgrad_workspace_bytes(B, H, H, ci, co)
    ws = torch.empty(nb // 4 + 16, device="cuda")
    assert lib.vc_conv3x3_wino_wgrad_f32(st(), B, H, H, ci, co, P(x), P(dy), P(dw), P(db), 0, P(ws), nb) == 0
    torch.cuda.synchronize()
    key = "%%dx%%dx%%d->%%d" %% (B, H, ci, co)
    out["wgrad " + key] = hashlib.sha256(dw.cpu().numpy().tobytes() + db.cpu().numpy().tobytes()).hexdigest()
print("RESULT " + json.dumps(out, sort_keys=True))
"""

STEP = r"""
import hashlib, json, sys, numpy as np, torch
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from gpu_util_step import tiny_finetune_trainer
tr = tiny_finetune_trainer()
for _ in range(2):
    tr.train_step()
torch.cuda.synchronize()
l = tr.losses()
out = {"losses": [float(np.float32(v)).hex() for v in l],
       "params": hashlib.sha256(tr.cap.store.p.cpu().numpy().tobytes()).hexdigest(),
       "vgg": hashlib.sha256(tr.vgg.store.p[:tr.vgg.o_fc].cpu().numpy().tobytes()).hexdigest()}
print("RESULT " + json.dumps(out, sort_keys=True))
"""


def _run(code, env_extra):
    env = dict(os.environ)
    for k in ("VC_WINO4_TG", "VC_WGRAD_XCD", "VC_LOGITS_DW", "VC_WINO4V"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_workgroup_orders_do_not_change_a_single_bit():
    code = KERNELS % ROOT
    ref = _run(code, {})
    assert len(ref) == 9
    for env in ({"VC_WINO4_TG": "0"}, {"VC_WINO4_TG": "2"}, {"VC_WINO4_TG": "8"}, {"VC_WGRAD_XCD": "0"}):
        got = _run(code, env)
        assert got == ref, (env, [k for k in ref if got.get(k) != ref[k]])


def test_chunk_width_one_keeps_conv1_1s_mask_bits_aligned_with_conv1_2s_data_gradient():
    """conv1_1's forward (csrc/conv_first.hip) writes conv1_2's ReLU mask at (block pair, channel tile); the F(4x4,3x3) data gradient
    used to index the mask by its chunk-remapped workgroup id, which equals that only while the chunk width covers both channel
    tiles of the 64-channel layer.  VC_WINO4_TG=1 is the order where the two differ: a whole fine-tune step (conv1_1's weight gradient
    flows through those bits) must stay bit-identical."""
    code = STEP % (ROOT, os.path.join(ROOT, "tests"))
    ref = _run(code, {})
    got = _run(code, {"VC_WINO4_TG": "1"})
    assert got == ref, (ref, got)


def test_issue_point_of_the_logits_weight_gradient_does_not_change_the_step():
    code = STEP % (ROOT, os.path.join(ROOT, "tests"))
    ref = _run(code, {})
    got = _run(code, {"VC_LOGITS_DW": "now"})
    assert got == ref, (ref, got)


def test_fine_tune_step_on_the_pre_transformed_convolutions_equals_the_fused_kernels_bit_for_bit():
    """round 6: with VC_WINO4V=1 (the default of the split-bf16 mode) conv4_1 .. conv5_3 forward and data gradient run
    vc_conv3x3_wino4v_* (the input transformed once per layer and pass); VC_WINO4V=0 keeps the fused F(4x4,3x3) kernel.  Same transforms, same MFMA order, same mask bits and routing codes: two whole
    fine-tune steps (losses, caption parameters, convolution parameters) must agree in every bit."""
    code = (STEP % (ROOT, os.path.join(ROOT, "tests"))).replace("tiny_finetune_trainer()", "tiny_finetune_trainer(4)")   # two images per chain: conv4_x qualifies too
    ref = _run(code, {"VC_WINO4V": "1"})
    got = _run(code, {"VC_WINO4V": "0"})
    assert got == ref, (ref, got)
