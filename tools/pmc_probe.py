"""Workload for SQ-counter probes (rocprofv3 --pmc ...): conv3_2-shaped forward / data gradient / weight gradient at
64 images -- the implicit-GEMM kernels of conv.hip and the patch-staged kernels of conv_patch.hip -- the conv4_2-shaped (4 x 7 K-tile) weight
gradient and the 8192^3 GEMM, three launches each.  Summarise with tools/pmc_probe_summary.py."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vae_captioning_amd import abi  # noqa: E402
from vae_captioning_amd.abi import ptr as P  # noqa: E402

lib = abi.load()
st = lambda: torch.cuda.current_stream().cuda_stream
B, H, ci, co = 64, 56, 256, 256
x = torch.rand(B, H, H, ci, device="cuda") * 2 - 1
w = torch.rand(3, 3, ci, co, device="cuda") * 2 - 1
bias = torch.rand(co, device="cuda")
y = torch.empty(B, H, H, co, device="cuda")
dy = torch.rand(B, H, H, co, device="cuda") * 2 - 1
dx = torch.empty(B, H, H, ci, device="cuda")
dw = torch.empty(3, 3, ci, co, device="cuda")
ws = torch.empty(max(lib.vc_conv3x3_wgrad_workspace_bytes(B, H, H, ci, co), lib.vc_conv3x3_wgrad_patch_workspace_bytes(B, H, H, ci, co),
                     lib.vc_conv3x3_wgrad_patch_workspace_bytes(B, 28, 28, 512, 512)) // 4 + 4, device="cuda")
wp, wpt = torch.empty(9 * ci * co, device="cuda"), torch.empty(9 * ci * co, device="cuda")
lib.vc_conv3x3_pack_f32(st(), ci, co, P(w), 0, P(wp))
lib.vc_conv3x3_pack_f32(st(), ci, co, P(w), 1, P(wpt))
tw = torch.empty(max(lib.vc_conv3x3_packed_workspace_bytes(B, H, H, ci, co, 0), 16) // 4 + 4, device="cuda")
# conv4_2 shape: the 4 x 7 K-tile weight-gradient kernel
x4 = torch.rand(B, 28, 28, 512, device="cuda") * 2 - 1
dy4 = torch.rand(B, 28, 28, 512, device="cuda") * 2 - 1
dw4 = torch.empty(3, 3, 512, 512, device="cuda")
A = torch.rand(8192, 8192, device="cuda") * 2 - 1
Bm = torch.rand(8192, 8192, device="cuda") * 2 - 1
C = torch.empty(8192, 8192, device="cuda")
vp, vpt = torch.empty(16 * ci * co, device="cuda"), torch.empty(16 * ci * co, device="cuda")
lib.vc_conv3x3_wino_pack_f32(st(), ci, co, P(w), 0, P(vp))
lib.vc_conv3x3_wino_pack_f32(st(), ci, co, P(w), 1, P(vpt))
wws = torch.empty(max(lib.vc_conv3x3_wino_wgrad_workspace_bytes(B, H, H, ci, co), lib.vc_conv3x3_wino_wgrad_workspace_bytes(B, 28, 28, 512, 512)) // 4 + 4, device="cuda")
for _ in range(3):
    # Winograd kernels (csrc/conv_wino.hip): conv3_2-shaped forward / data gradient / weight gradient, conv4_2-shaped weight gradient
    lib.vc_conv3x3_wino_fwd_f32(st(), B, H, H, ci, co, P(x), P(vp), P(bias), P(y), None, 1)
    lib.vc_conv3x3_wino_dgrad_f32(st(), B, H, H, ci, co, P(dy), P(vpt), P(x), P(dx))
    lib.vc_conv3x3_wino_wgrad_f32(st(), B, H, H, ci, co, P(x), P(dy), P(dw), None, 0, P(wws), wws.numel() * 4)
    lib.vc_conv3x3_wino_wgrad_f32(st(), B, 28, 28, 512, 512, P(x4), P(dy4), P(dw4), None, 0, P(wws), wws.numel() * 4)
    lib.vc_conv3x3_fwd_f32(st(), B, H, H, ci, co, P(x), P(w), P(bias), P(y), 1, None, 0)
    lib.vc_conv3x3_dgrad_f32(st(), B, H, H, ci, co, P(dy), P(w), P(x), P(dx), None, 0)
    lib.vc_conv3x3_wgrad_f32(st(), B, H, H, ci, co, P(x), P(dy), P(dw), None, 0, P(ws), ws.numel() * 4)
    lib.vc_conv3x3_fwd_packed_f32(st(), B, H, H, ci, co, P(x), P(wp), P(bias), P(y), 1, P(tw), tw.numel() * 4)
    lib.vc_conv3x3_dgrad_packed_f32(st(), B, H, H, ci, co, P(dy), P(wpt), P(x), P(dx), P(tw), tw.numel() * 4)
    lib.vc_conv3x3_wgrad_patch_f32(st(), B, H, H, ci, co, P(x), P(dy), P(dw), None, 0, P(ws), ws.numel() * 4)
    lib.vc_conv3x3_wgrad_patch_f32(st(), B, 28, 28, 512, 512, P(x4), P(dy4), P(dw4), None, 0, P(ws), ws.numel() * 4)
    lib.vc_gemm_f32(st(), 0, 0, 8192, 8192, 8192, P(A), 8192, P(Bm), 8192, P(C), 8192, None, 0, None, 0)
torch.cuda.synchronize()
