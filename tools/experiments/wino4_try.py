"""HISTORY.md section 4g: the Winograd F(4x4,3x3) forward / data gradient of csrc/conv_wino4.hip against the
library's F(2x2,3x3) kernel -- values and time on the VGG16 layer shapes.  Run: python tools/experiments/wino4_try.py [images] [dgrad|bits]; ablations: make -C vae_captioning_amd/csrc wino4abl W4FLAGS=-DW4_ABL=n, then
VC_LIB=vae_captioning_amd/lib/libvaecap_wino4abl.so
(W4_ONLY=conv2_2,conv4_2 restricts the layers)."""
import sys, torch
sys.path.insert(0, ".")
from vae_captioning_amd import abi
from vae_captioning_amd.abi import ptr as P
import os
lib = abi.load(os.environ.get("VC_LIB"))


def w4(name, *a):
    return getattr(lib, name)(*a)
st = lambda: torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dg = len(sys.argv) > 2 and sys.argv[2] in ("dgrad", "bits")
bits = len(sys.argv) > 2 and sys.argv[2] == "bits"   # data gradient with the ReLU mask as bits (what the training step runs)
if os.environ.get("VC_LIB"):
    pass
shapes = [("conv1_2", 224, 64, 64), ("conv2_1", 112, 64, 128), ("conv2_2", 112, 128, 128), ("conv3_1", 56, 128, 256), ("conv3_2", 56, 256, 256),
          ("conv4_1", 28, 256, 512), ("conv4_2", 28, 512, 512), ("conv5_2", 14, 512, 512)]
if os.environ.get("W4_ONLY"):
    shapes = [s for s in shapes if s[0] in os.environ["W4_ONLY"].split(",")]
if B < 8:
    shapes = [("s16", 16, 8, 32), ("s20", 20, 16, 32), ("s28", 28, 32, 64)] + shapes
tot2 = tot4 = 0.0
for name, H, ci, co in shapes:
    torch.manual_seed(1)
    x = torch.relu(torch.randn(B, H, H, co if dg else ci, device="cuda"))
    w = torch.randn(3, 3, ci, co, device="cuda") * (2.0 / (9 * ci)) ** 0.5
    bias = torch.randn(co, device="cuda")
    src = torch.randn(B, H, H, ci, device="cuda")
    no = ci if dg else co
    y2 = torch.empty(B, H, H, no, device="cuda"); y4 = torch.empty_like(y2)
    wp2 = torch.empty(16 * ci * co, device="cuda"); wp4 = torch.empty(36 * ci * co, device="cuda")
    lib.vc_conv3x3_wino_pack_f32(st(), ci, co, P(w), int(dg), P(wp2))
    w4("vc_conv3x3_wino4_pack_f32", st(), ci, co, P(w), int(dg), P(wp4))
    if bits:
        m2 = torch.zeros(lib.vc_conv3x3_wino_mask_words(B, H, H, ci), dtype=torch.int32, device="cuda")
        m4 = torch.zeros(lib.vc_conv3x3_wino4_mask_words(B, H, H, ci), dtype=torch.int32, device="cuda")
        m2.random_(0, 2 ** 31 - 1); m4.random_(0, 2 ** 31 - 1)
        f2 = lambda: lib.vc_conv3x3_wino_dgrad_bits_f32(st(), B, H, H, ci, co, P(x), P(wp2), P(m2), P(y2))
        f4 = lambda: lib.vc_conv3x3_wino4_dgrad_bits_f32(st(), B, H, H, ci, co, P(x), P(wp4), P(m4), P(y4))
    elif dg:
        f2 = lambda: lib.vc_conv3x3_wino_dgrad_f32(st(), B, H, H, ci, co, P(x), P(wp2), P(src), P(y2))
        f4 = lambda: w4("vc_conv3x3_wino4_dgrad_f32", st(), B, H, H, ci, co, P(x), P(wp4), P(src), P(y4))
    else:
        f2 = lambda: lib.vc_conv3x3_wino_fwd_f32(st(), B, H, H, ci, co, P(x), P(wp2), P(bias), P(y2), None, 1)
        f4 = lambda: w4("vc_conv3x3_wino4_fwd_f32", st(), B, H, H, ci, co, P(x), P(wp4), P(bias), P(y4), None, 1)
    f2(); f4(); torch.cuda.synchronize()
    err = float((y2 - y4).abs().max()); scale = float(y2.abs().max())
    ts = []
    for f in (f2, f4):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * B * H * H * 9 * ci * co
    tot2 += ts[0]; tot4 += ts[1]
    print("%-8s %3d %3d->%3d  max|d| %.2e of %.2f  F(2,3) %.3f ms %5.0f TF  F(4,3) %.3f ms %5.0f TF  x%.2f" % (
        name, H, ci, co, err, scale, ts[0], fl / ts[0] / 1e9, ts[1], fl / ts[1] / 1e9, ts[0] / ts[1]))
print("sum F(2,3) %.3f ms, F(4,3) %.3f ms" % (tot2, tot4))
