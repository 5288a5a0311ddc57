"""Workload for SQ-counter probes of the Winograd forward / data-gradient kernel only (tools/sq_probe.sh with PROBE=tools/pmc_probe_wino.py):
conv3_2-, conv1_2-, conv4_2- and conv5_2-shaped forward + the conv3_2 data gradient at 64 images, three launches each.
FAMILY=4 in the environment: the F(4x4,3x3) entries (vc_conv3x3_wino4_*) instead of the F(2x2,3x3) ones."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vae_captioning_amd import abi  # noqa: E402
from vae_captioning_amd.abi import ptr as P  # noqa: E402

lib = abi.load()
F4 = os.environ.get("FAMILY", "2") in ("4", "4v")
F4V = os.environ.get("FAMILY", "2") == "4v"   # round 6: the pre-transformed form (vc_conv3x3_wino4v_*) where it takes the shape, the fused one elsewhere
PRE = "vc_conv3x3_wino4_" if F4 else "vc_conv3x3_wino_"
fn = lambda e: getattr(lib, PRE + e)
st = lambda: torch.cuda.current_stream().cuda_stream
B = 64
cases = []
for (H, ci, co) in ((56, 256, 256), (224, 64, 64), (28, 512, 512), (14, 512, 512)):
    x = torch.rand(B, H, H, ci, device="cuda") * 2 - 1
    w = torch.rand(3, 3, ci, co, device="cuda") * 2 - 1
    bias = torch.rand(co, device="cuda")
    y = torch.empty(B, H, H, co, device="cuda")
    vp, vpt = torch.empty((36 if F4 else 16) * ci * co, device="cuda"), torch.empty((36 if F4 else 16) * ci * co, device="cuda")
    fn("pack_f32")(st(), ci, co, P(w), 0, P(vp))
    fn("pack_f32")(st(), ci, co, P(w), 1, P(vpt))
    cases.append((H, ci, co, x, bias, y, vp, vpt))
for _ in range(3):
    for i, (H, ci, co, x, bias, y, vp, vpt) in enumerate(cases):
        if F4V and lib.vc_conv3x3_wino4v_supported(B, H, H, ci, co, 0):
            nb = lib.vc_conv3x3_wino4v_workspace_bytes(B, H, H, max(ci, co))
            vws = torch.empty(nb // 4, device="cuda")
            lib.vc_conv3x3_wino4v_fwd_f32(st(), B, H, H, ci, co, P(x), P(vp), P(bias), P(y), None, 1, P(vws), nb)
            if i == 0:
                lib.vc_conv3x3_wino4v_dgrad_f32(st(), B, H, H, ci, co, P(y), P(vpt), P(x), P(x), P(vws), nb)
            continue
        fn("fwd_f32")(st(), B, H, H, ci, co, P(x), P(vp), P(bias), P(y), None, 1)
        if i == 0:
            fn("dgrad_f32")(st(), B, H, H, ci, co, P(y), P(vpt), P(x), P(x))
torch.cuda.synchronize()
