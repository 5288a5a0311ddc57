"""Round 6: the F(4x4,3x3) convolution with a pre-transformed input (vc_conv3x3_wino4v_*: csrc/conv_wino4.hip MODE 2) against the fused
kernel (vc_conv3x3_wino4_*) -- outputs, mask bits, pooled outputs and routing codes bit for bit, and time -- on the VGG16 layer shapes it
takes.  Run: python tools/experiments/wino4v_try.py [images]  (W4_ONLY=conv4_2,... restricts the layers)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_captioning_amd import abi
from vae_captioning_amd.abi import ptr as P

lib = abi.load(os.environ.get("VC_LIB"))
st = lambda: torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [("conv3_1", 56, 128, 256), ("conv3_2", 56, 256, 256), ("conv4_1", 28, 256, 512), ("conv4_2", 28, 512, 512), ("conv5_2", 14, 512, 512)]
if os.environ.get("W4_ONLY"):
    shapes = [s for s in shapes if s[0] in os.environ["W4_ONLY"].split(",")]
if B < 8:
    shapes = [("s28", 28, 32, 64), ("s14", 14, 8, 32), ("s30", 30, 16, 32)] + shapes


def timeit(f):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


tot = {}
for name, H, ci, co in shapes:
    torch.manual_seed(1)
    w = torch.randn(3, 3, ci, co, device="cuda") * (2.0 / (9 * ci)) ** 0.5
    bias = torch.randn(co, device="cuda")
    wp, wpt = torch.empty(36 * ci * co, device="cuda"), torch.empty(36 * ci * co, device="cuda")
    lib.vc_conv3x3_wino4_pack_f32(st(), ci, co, P(w), 0, P(wp))
    lib.vc_conv3x3_wino4_pack_f32(st(), ci, co, P(w), 1, P(wpt))
    x = torch.relu(torch.randn(B, ci // 4, H, H, 4, device="cuda"))
    dy = torch.randn(B, co // 4, H, H, 4, device="cuda")
    for kind in ("fwd_mask", "fwd_pool", "dgrad_bits", "dgrad"):
        dg = kind.startswith("dgrad")
        if not lib.vc_conv3x3_wino4v_supported(B, H, H, ci, co, int(dg)):
            print("%-8s %-10s not supported by wino4v" % (name, kind))
            continue
        vws = torch.empty(lib.vc_conv3x3_wino4v_workspace_bytes(B, H, H, co if dg else ci) // 4, device="cuda")
        nb = vws.numel() * 4
        out = [torch.zeros(B, (ci if dg else co) // 4, H, H, 4, device="cuda") for _ in range(2)]
        mk = [torch.zeros(lib.vc_conv3x3_wino4_mask_words(B, H, H, co), dtype=torch.int32, device="cuda") for _ in range(2)]
        pl = [torch.zeros(B, co // 4, H // 2, H // 2, 4, device="cuda") for _ in range(2)]
        pb = [torch.zeros(lib.vc_conv3x3_wino_pool_words(B, H, H, co), dtype=torch.int32, device="cuda") for _ in range(2)]
        mi = torch.zeros(lib.vc_conv3x3_wino4_mask_words(B, H, H, ci), dtype=torch.int32, device="cuda").random_(0, 2 ** 31 - 1)
        if kind == "fwd_mask":
            f = [lambda: lib.vc_conv3x3_wino4_fwd_mask_f32(st(), B, H, H, ci, co, P(x), P(wp), P(bias), P(out[0]), 1, P(mk[0])),
                 lambda: lib.vc_conv3x3_wino4v_fwd_mask_f32(st(), B, H, H, ci, co, P(x), P(wp), P(bias), P(out[1]), 1, P(mk[1]), P(vws), nb)]
        elif kind == "fwd_pool":
            f = [lambda: lib.vc_conv3x3_wino4_fwd_pool_f32(st(), B, H, H, ci, co, P(x), P(wp), P(bias), P(out[0]), P(pl[0]), P(pb[0])),
                 lambda: lib.vc_conv3x3_wino4v_fwd_pool_f32(st(), B, H, H, ci, co, P(x), P(wp), P(bias), P(out[1]), P(pl[1]), P(pb[1]), P(vws), nb)]
        elif kind == "dgrad_bits":
            f = [lambda: lib.vc_conv3x3_wino4_dgrad_bits_f32(st(), B, H, H, ci, co, P(dy), P(wpt), P(mi), P(out[0])),
                 lambda: lib.vc_conv3x3_wino4v_dgrad_bits_f32(st(), B, H, H, ci, co, P(dy), P(wpt), P(mi), P(out[1]), P(vws), nb)]
        else:
            f = [lambda: lib.vc_conv3x3_wino4_dgrad_f32(st(), B, H, H, ci, co, P(dy), P(wpt), P(x), P(out[0])),
                 lambda: lib.vc_conv3x3_wino4v_dgrad_f32(st(), B, H, H, ci, co, P(dy), P(wpt), P(x), P(out[1]), P(vws), nb)]
        f[0](); f[1](); torch.cuda.synchronize()
        same = torch.equal(out[0], out[1]) and torch.equal(mk[0], mk[1]) and torch.equal(pl[0], pl[1]) and torch.equal(pb[0], pb[1])
        err = float((out[0] - out[1]).abs().max())
        t = [timeit(g) for g in f]
        fl = 2.0 * B * H * H * 9 * ci * co
        tot.setdefault(kind, [0.0, 0.0])
        mult = {"conv3_2": 2, "conv4_2": 2, "conv5_2": 3}.get(name, 1)
        tot[kind][0] += mult * t[0]; tot[kind][1] += mult * t[1]
        print("%-8s %-10s B=%d %3d %3d->%3d  %s (max|d| %.1e of %.2f)  fused %.3f ms %4.0f TF | pre-transformed %.3f ms %4.0f TF  x%.2f" % (
            name, kind, B, H, ci, co, "bit-identical" if same else "DIFFERENT", err, float(out[0].abs().max()), t[0], fl / t[0] / 1e9, t[1], fl / t[1] / 1e9, t[0] / t[1]), flush=True)
for k, v in tot.items():
    print("sum over the layers (conv3_2, conv4_2 x2, conv5_2 x3) %-10s fused %.3f ms, pre-transformed %.3f ms  x%.2f" % (k, v[0], v[1], v[0] / v[1]))
