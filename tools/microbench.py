#!/usr/bin/env python
"""Kernel microbenchmarks on one MI355X (HIP events, within-process interleaved rounds):
   python tools/microbench.py gemm | ablate | conv | lstm"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_captioning_amd import abi  # noqa: E402
from vae_captioning_amd.abi import ptr as P  # noqa: E402

lib = abi.load(os.environ.get("VC_LIB"))   # VC_LIB: an ablation build (make -C vae_captioning_amd/csrc ablate)


def st():
    return torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=10, rounds=3):
    fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps)
    return float(np.median(best)), float(min(best))


def rnd(*shape):
    return torch.rand(*shape, device="cuda") * 2 - 1


def gemm():
    for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (25600, 10000, 512), (6400, 10000, 512), (28160, 2048, 256), (200704, 256, 2304)]:
        A, B, C = rnd(M, K), rnd(K, N), torch.empty(M, N, device="cuda")
        ws = torch.empty(max(lib.vc_gemm_workspace_bytes(M, N, K), 16) // 4 + 4, device="cuda")
        med, mn = timeit(lambda: lib.vc_gemm_f32(st(), 0, 0, M, N, K, P(A), K, P(B), N, P(C), N, None, 0, P(ws), ws.numel() * 4))
        print("gemm NN %6d x %6d x %6d: %8.3f ms  %6.1f TFLOP/s (best %.1f)" % (M, N, K, med, 2e-9 * M * N * K / med, 2e-9 * M * N * K / mn))


def gemmx():
    """f32 MFMA against split-bf16 (bf16x3) on the training products: logits forward / data gradient / weight gradient at cfg2
    (27520 rows) and cfg4 (6880 rows), LSTM projections, fc1, square references."""
    for (ta, tb, M, N, K) in [(0, 0, 27520, 10000, 512), (0, 1, 27520, 512, 10000), (1, 0, 512, 10000, 27520),
                              (0, 0, 6880, 10000, 512), (0, 1, 6880, 512, 10000), (1, 0, 512, 10000, 6880),
                              (0, 0, 28160, 2048, 256), (0, 1, 28160, 256, 2048), (1, 0, 256, 2048, 28160), (1, 0, 512, 2048, 28160),
                              (0, 0, 64, 4096, 25088), (0, 1, 64, 25088, 4096), (1, 0, 25088, 4096, 64),
                              (0, 0, 4096, 4096, 4096), (0, 0, 8192, 8192, 8192)]:
        A = rnd(K, M) if ta else rnd(M, K)
        B = rnd(N, K) if tb else rnd(K, N)
        C = torch.empty(M, N, device="cuda")
        ws = torch.empty(max(lib.vc_gemm_workspace_bytes(M, N, K), 16) // 4 + 4, device="cuda")
        res = []
        for fn in (lib.vc_gemm_f32, lib.vc_gemm_bf16x3_f32):
            med, mn = timeit(lambda: fn(st(), ta, tb, M, N, K, P(A), M if ta else K, P(B), K if tb else N, P(C), N, None, 0, P(ws), ws.numel() * 4))
            res.append(med)
        fl = 2e-9 * M * N * K
        print("gemm ta=%d tb=%d %6d x %6d x %6d: f32 %7.3f ms %6.1f TF | bf16x3 %7.3f ms %6.1f TF effective = %6.1f TF on the bf16 pipe (%.3f of 2500)  x%.2f"
              % (ta, tb, M, N, K, res[0], fl / res[0], res[1], fl / res[1], 3 * fl / res[1], 3 * fl / res[1] / 2500, res[0] / res[1]), flush=True)


def gemmshape():
    """one product, VC_SHAPE="ta,tb,M,N,K" (default: the AG / GMM heads of cfg3, [1280, 512] x [512, 27000], vae_model/encoder.py:90-107):
    the launch tools/kernel_pmc.sh counts when it is asked about a single GEMM of a step"""
    ta, tb, M, N, K = [int(v) for v in os.environ.get("VC_SHAPE", "0,0,1280,27000,512").split(",")]
    A = rnd(K, M) if ta else rnd(M, K)
    B = rnd(N, K) if tb else rnd(K, N)
    C = torch.empty(M, N, device="cuda")
    ws = torch.empty(max(lib.vc_gemm_workspace_bytes(M, N, K), 16) // 4 + 4, device="cuda")
    fl = 4 if os.environ.get("VC_PRECISION") == "bf16x3" else 0
    med, mn = timeit(lambda: lib.vc_gemm_f32(st(), ta, tb, M, N, K, P(A), M if ta else K, P(B), K if tb else N, P(C), N, None, fl, P(ws), ws.numel() * 4))
    print("gemm ta=%d tb=%d %d x %d x %d: %.3f ms  %.1f TFLOP/s" % (ta, tb, M, N, K, med, 2e-9 * M * N * K / med))


def conv1():
    """conv1_1's own kernels (csrc/conv_first.hip) against the general 3x3 kernels on the zero-padded 4-channel form; both are
    HBM-bound on the [B,224,224,64] activation (822 MB at B = 64: ~0.14 ms at 6.3 TB/s)"""
    B, H = 64, 224
    x4 = rnd(B, H, H, 4)
    w, bias = rnd(3, 3, 3, 64), rnd(64)
    w4 = torch.zeros(3, 3, 4, 64, device="cuda")
    w4[:, :, :3] = w
    y, dy = torch.empty(B, H, H, 64, device="cuda"), rnd(B, H, H, 64)
    dw, db, dw4 = torch.empty(3, 3, 3, 64, device="cuda"), torch.empty(64, device="cuda"), torch.empty(3, 3, 4, 64, device="cuda")
    ws = torch.empty(max(lib.vc_conv1_wgrad_workspace_bytes(), lib.vc_conv3x3_wgrad_workspace_bytes(B, H, H, 4, 64)) // 4 + 4, device="cuda")
    gb = 1e-9 * B * H * H * (64 + 4) * 4
    for nm, fn in (("conv1 fwd", lambda: lib.vc_conv1_fwd_f32(st(), B, H, H, P(x4), P(w), P(bias), P(y), 1)),
                   ("3x3 fwd (Cin 4)", lambda: lib.vc_conv3x3_fwd_f32(st(), B, H, H, 4, 64, P(x4), P(w4), P(bias), P(y), 1, None, 0)),
                   ("conv1 wgrad", lambda: lib.vc_conv1_wgrad_f32(st(), B, H, H, P(x4), P(dy), P(dw), P(db), 0, P(ws), ws.numel() * 4)),
                   ("3x3 wgrad (Cin 4)", lambda: lib.vc_conv3x3_wgrad_f32(st(), B, H, H, 4, 64, P(x4), P(dy), P(dw4), P(db), 0, P(ws), ws.numel() * 4))):
        med, mn = timeit(fn, reps=10)
        print("%-18s B=%d: %7.3f ms  %5.2f TB/s (activation once + input once)" % (nm, B, med, gb / med), flush=True)


def gemmtrain():
    """the tb = 0 products of a training step (cfg4: 6880 rows, cfg2: 27520 rows).  Round 2 measured a register-B variant on these
    (B rows straight to registers as the operand of four interleaved MFMA tiles, 128 x 256 tiles, two workgroups per CU):
    116.5 / 107.2 / 118.0 TFLOP/s against 119.6 / 119.4 / 124.9 of this kernel on 27520x10000x512 / 27520x2048x512 /
    512x10000x27520 (only 8192^3 gained, 136.6 vs 133.3) -- per-tile prologue / epilogue with 16 K-tiles, not the main loop,
    bounds these shapes, and three to four resident workgroups hide it better than two; not adopted."""
    for (ta, M, N, K) in [(0, 6880, 10000, 512), (0, 27520, 10000, 512), (0, 6880, 2048, 512), (0, 27520, 2048, 512), (0, 6880, 2048, 256),
                          (1, 512, 10000, 6880), (1, 512, 10000, 27520), (1, 512, 2048, 6880), (1, 512, 2048, 27520), (1, 256, 2048, 27520),
                          (0, 64, 4096, 25088), (1, 25088, 4096, 64), (0, 64, 4096, 4096), (0, 8192, 8192, 8192)]:
        A = rnd(K, M) if ta else rnd(M, K)
        B = rnd(K, N)
        C = torch.empty(M, N, device="cuda")
        ws = torch.empty(max(lib.vc_gemm_workspace_bytes(M, N, K), 16) // 4 + 4, device="cuda")
        med, mn = timeit(lambda: lib.vc_gemm_f32(st(), ta, 0, M, N, K, P(A), M if ta else K, P(B), N, P(C), N, None, 0, P(ws), ws.numel() * 4), reps=5)
        print("gemm ta=%d tb=0 %6d x %6d x %6d: %8.3f ms  %6.1f TFLOP/s" % (ta, M, N, K, med, 2e-9 * M * N * K / med), flush=True)


def gemmt():
    """transposed-operand forms at conv-wgrad-like and dense-backward shapes"""
    for (ta, tb, M, N, K) in [(0, 0, 8192, 8192, 8192), (0, 1, 8192, 8192, 8192), (1, 0, 8192, 8192, 8192), (1, 1, 8192, 8192, 8192), (0, 0, 2304, 256, 200704), (1, 0, 2304, 256, 200704), (0, 1, 200704, 256, 2304), (1, 0, 512, 10000, 25600), (0, 1, 25600, 512, 10000)]:
        A = rnd(K, M) if ta else rnd(M, K)
        B = rnd(N, K) if tb else rnd(K, N)
        C = torch.empty(M, N, device="cuda")
        ws = torch.empty(max(lib.vc_gemm_workspace_bytes(M, N, K), 16) // 4 + 4, device="cuda")
        med, mn = timeit(lambda: lib.vc_gemm_f32(st(), ta, tb, M, N, K, P(A), M if ta else K, P(B), K if tb else N, P(C), N, None, 0, P(ws), ws.numel() * 4), reps=5)
        print("gemm ta=%d tb=%d %6d x %6d x %6d: %8.3f ms  %6.1f TFLOP/s" % (ta, tb, M, N, K, med, 2e-9 * M * N * K / med))


def ablate(shapes=((4096, 4096, 4096), (8192, 8192, 8192), (25600, 10000, 512), (200704, 256, 2304), (12544, 512, 4608))):
    import ctypes
    mb_path = os.path.join(os.path.dirname(abi.LIB_PATH), "libvaecap_microbench.so")  # make -C vae_captioning_amd/csrc microbench
    mb = ctypes.CDLL(mb_path)
    mb.vc_debug_gemm_ablate_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    names = {0: "full", 8: "double-buffered", 2: "no lds-store/barrier", 7: "mfma only"}
    for (M, N, K) in shapes:
        M, N = M // 128 * 128, N // 128 * 128
        A, B, C = rnd(M, K), rnd(K, N), torch.empty(M, N, device="cuda")
        ref = None
        for v, nm in names.items():
            med, mn = timeit(lambda: mb.vc_debug_gemm_ablate_f32(st(), v, M, N, K, P(A), P(B), P(C)))
            chk = ""
            if v == 0:
                ref = C.clone()
            elif v == 8:
                chk = " maxdiff vs full %.3g" % float((C - ref).abs().max())
            print("ablate %dx%dx%d %d %-22s %8.3f ms  %6.1f TFLOP/s%s" % (M, N, K, v, nm, med, 2e-9 * M * N * K / med, chk))


def conv():
    B = 64
    for (name, H, ci, co) in [("1_1", 224, 4, 64), ("1_2", 224, 64, 64), ("2_1", 112, 64, 128), ("2_2", 112, 128, 128), ("3_1", 56, 128, 256), ("3_2", 56, 256, 256), ("4_1", 28, 256, 512), ("4_2", 28, 512, 512), ("5_2", 14, 512, 512)]:
        x, w, bias = rnd(B, H, H, ci), rnd(3, 3, ci, co), rnd(co)
        y, dx, dw = torch.empty(B, H, H, co, device="cuda"), torch.empty(B, H, H, ci, device="cuda"), torch.empty(3, 3, ci, co, device="cuda")
        dy = rnd(B, H, H, co)
        ws = torch.empty(lib.vc_conv3x3_wgrad_workspace_bytes(B, H, H, ci, co) // 4 + 4, device="cuda")
        tw = torch.empty(max(lib.vc_conv3x3_fwd_workspace_bytes(B, H, H, ci, co), lib.vc_conv3x3_dgrad_workspace_bytes(B, H, H, ci, co), 16) // 4 + 4, device="cuda")
        tb = tw.numel() * 4 if os.environ.get("VC_NO_TAIL") != "1" else 0
        fl = 2e-9 * B * H * H * 9 * ci * co
        for nm, fn in (("fwd", lambda: lib.vc_conv3x3_fwd_f32(st(), B, H, H, ci, co, P(x), P(w), P(bias), P(y), 1, P(tw), tb)),
                       ("dgrad", lambda: lib.vc_conv3x3_dgrad_f32(st(), B, H, H, ci, co, P(dy), P(w), P(x), P(dx), P(tw), tb)),
                       ("dgr-nomask", lambda: lib.vc_conv3x3_dgrad_f32(st(), B, H, H, ci, co, P(dy), P(w), None, P(dx), P(tw), tb)),
                       ("wgrad", lambda: lib.vc_conv3x3_wgrad_f32(st(), B, H, H, ci, co, P(x), P(dy), P(dw), P(bias), 0, P(ws), ws.numel() * 4)),
                       ("wgrad-nobias", lambda: lib.vc_conv3x3_wgrad_f32(st(), B, H, H, ci, co, P(x), P(dy), P(dw), None, 0, P(ws), ws.numel() * 4))):
            med, mn = timeit(fn, reps=5)
            print("conv%s %-10s B=%d H=%3d %3d->%3d: %8.3f ms  %6.1f TFLOP/s" % (name, nm, B, H, ci, co, med, fl / med))


def winoab():
    """Winograd forward / data gradient only, VGG16 layer shapes at 64 and 32 images (round 3 A/B: profiles/r03_wino_fwd_dgrad_round2_kernel.txt keeps the round-2 32x32x2 kernel's
    numbers from the same program); algorithmic TFLOP/s"""
    tot = {}
    for B in (64, 32):
        for (name, H, ci, co) in [("1_2", 224, 64, 64), ("2_1", 112, 64, 128), ("2_2", 112, 128, 128), ("3_1", 56, 128, 256), ("3_2", 56, 256, 256),
                                  ("4_1", 28, 256, 512), ("4_2", 28, 512, 512), ("5_2", 14, 512, 512)]:
            x, w, bias = rnd(B, H, H, ci), rnd(3, 3, ci, co), rnd(co)
            y, dx = torch.empty(B, H, H, co, device="cuda"), torch.empty(B, H, H, ci, device="cuda")
            dy = rnd(B, H, H, co)
            vp, vpt = torch.empty(16 * ci * co, device="cuda"), torch.empty(16 * ci * co, device="cuda")
            lib.vc_conv3x3_wino_pack_f32(st(), ci, co, P(w), 0, P(vp))
            lib.vc_conv3x3_wino_pack_f32(st(), ci, co, P(w), 1, P(vpt))
            bits = torch.zeros(lib.vc_conv3x3_wino_mask_words(B, H, H, ci), dtype=torch.int32, device="cuda")
            fl = 2e-9 * B * H * H * 9 * ci * co
            for nm, fn in (("fwd", lambda: lib.vc_conv3x3_wino_fwd_f32(st(), B, H, H, ci, co, P(x), P(vp), P(bias), P(y), None, 1)),
                           ("dgrad", lambda: lib.vc_conv3x3_wino_dgrad_f32(st(), B, H, H, ci, co, P(dy), P(vpt), P(x), P(dx))),
                           ("dgrad-bits", lambda: lib.vc_conv3x3_wino_dgrad_bits_f32(st(), B, H, H, ci, co, P(dy), P(vpt), P(bits), P(dx)))):
                med, mn = timeit(fn, reps=5)
                tot[(B, nm)] = tot.get((B, nm), 0.0) + med * {"1_2": 1, "2_1": 1, "2_2": 1, "3_1": 1, "3_2": 2, "4_1": 1, "4_2": 2, "5_2": 3}[name]
                print("conv%s %-10s B=%d H=%3d %3d->%3d: %8.3f ms  %6.1f TFLOP/s" % (name, nm, B, H, ci, co, med, fl / med), flush=True)
    for k in sorted(tot):
        print("sum over the twelve layers B=%d %-10s %8.3f ms" % (k[0], k[1], tot[k]))


def winoq():
    """quick form of winoab: conv3_2 / conv1_2 / conv4_2 / conv5_2 forward at 64 images (ablation builds: VC_LIB=.../libvaecap_ablN.so)"""
    B = 64
    for (name, H, ci, co) in [("3_2", 56, 256, 256), ("1_2", 224, 64, 64), ("4_2", 28, 512, 512), ("5_2", 14, 512, 512)]:
        x, w, bias = rnd(B, H, H, ci), rnd(3, 3, ci, co), rnd(co)
        y = torch.empty(B, H, H, co, device="cuda")
        vp = torch.empty(16 * ci * co, device="cuda")
        lib.vc_conv3x3_wino_pack_f32(st(), ci, co, P(w), 0, P(vp))
        fl = 2e-9 * B * H * H * 9 * ci * co
        med, mn = timeit(lambda: lib.vc_conv3x3_wino_fwd_f32(st(), B, H, H, ci, co, P(x), P(vp), P(bias), P(y), None, 1), reps=5)
        print("%s conv%s fwd: %8.3f ms  %6.1f TFLOP/s" % (os.environ.get("VC_LIB", "default")[-12:], name, med, fl / med), flush=True)


def winowq():
    """quick form of winow: Winograd weight gradient only (ablation builds: VC_LIB=.../libvaecap_wgablN.so)"""
    B = 64
    for (name, H, ci, co) in [("1_2", 224, 64, 64), ("2_2", 112, 128, 128), ("3_2", 56, 256, 256), ("4_2", 28, 512, 512), ("5_2", 14, 512, 512)]:
        x, dy, bias = rnd(B, H, H, ci), rnd(B, H, H, co), rnd(co)
        dw = torch.empty(3, 3, ci, co, device="cuda")
        ws = torch.empty(lib.vc_conv3x3_wino_wgrad_workspace_bytes(B, H, H, ci, co) // 4 + 4, device="cuda")
        fl = 2e-9 * B * H * H * 9 * ci * co
        med, mn = timeit(lambda: lib.vc_conv3x3_wino_wgrad_f32(st(), B, H, H, ci, co, P(x), P(dy), P(dw), P(bias), 0, P(ws), ws.numel() * 4), reps=5)
        print("%s conv%s wgrad: %8.3f ms  %6.1f TFLOP/s" % (os.environ.get("VC_LIB", "default")[-14:], name, med, fl / med), flush=True)


def winow():
    """Winograd F(3x3,2x2) weight gradient, VGG16 layer shapes at 64 images (algorithmic TFLOP/s); the last line sums the twelve layers"""
    B = 64
    tot = 0.0
    for (name, H, ci, co) in [("1_2", 224, 64, 64), ("2_1", 112, 64, 128), ("2_2", 112, 128, 128), ("3_1", 56, 128, 256), ("3_2", 56, 256, 256),
                              ("4_1", 28, 256, 512), ("4_2", 28, 512, 512), ("5_2", 14, 512, 512)]:
        x, dy, bias = rnd(B, H, H, ci), rnd(B, H, H, co), rnd(co)
        dw = torch.empty(3, 3, ci, co, device="cuda")
        ws = torch.empty(lib.vc_conv3x3_wino_wgrad_workspace_bytes(B, H, H, ci, co) // 4 + 4, device="cuda")
        fl = 2e-9 * B * H * H * 9 * ci * co
        med, mn = timeit(lambda: lib.vc_conv3x3_wino_wgrad_f32(st(), B, H, H, ci, co, P(x), P(dy), P(dw), P(bias), 0, P(ws), ws.numel() * 4), reps=5)
        tot += med * {"1_2": 1, "2_1": 1, "2_2": 1, "3_1": 1, "3_2": 2, "4_1": 1, "4_2": 2, "5_2": 3}[name]
        print("conv%s %-12s B=%d H=%3d %3d->%3d: %8.3f ms  %6.1f TFLOP/s" % (name, "wgrad-wino", B, H, ci, co, med, fl / med), flush=True)
    print("sum over the twelve layers B=%d wgrad %8.3f ms" % (B, tot))


def wgbx():
    """direct split-bf16 weight gradient (conv_wgrad_bx.hip) beside the f32 Winograd F(3x3,2x2) kernel, VGG16 layer shapes at VC_WG_B images
    (default 64; algorithmic TFLOP/s); the last line sums the twelve layers.  Ablation builds: VC_LIB=.../libvaecap_wbablN.so"""
    B = int(os.environ.get("VC_WG_B", "64"))
    tot = [0.0, 0.0]
    only = os.environ.get("VC_WG_ONLY")   # e.g. 3_2: one layer (counter passes)
    for (name, H, ci, co) in [("1_2", 224, 64, 64), ("2_1", 112, 64, 128), ("2_2", 112, 128, 128), ("3_1", 56, 128, 256), ("3_2", 56, 256, 256),
                              ("4_1", 28, 256, 512), ("4_2", 28, 512, 512), ("5_2", 14, 512, 512)]:
        if only and name != only:
            continue
        x, dy, bias = rnd(B, H, H, ci), rnd(B, H, H, co), rnd(co)
        dw = torch.empty(3, 3, ci, co, device="cuda")
        ws = torch.empty(max(lib.vc_conv3x3_wino_wgrad_workspace_bytes(B, H, H, ci, co), lib.vc_conv3x3_bx_wgrad_workspace_bytes(B, H, H, ci, co)) // 4 + 4, device="cuda")
        fl = 2e-9 * B * H * H * 9 * ci * co
        mult = {"1_2": 1, "2_1": 1, "2_2": 1, "3_1": 1, "3_2": 2, "4_1": 1, "4_2": 2, "5_2": 3}[name]
        mb, _ = timeit(lambda: lib.vc_conv3x3_bx_wgrad_f32(st(), B, H, H, ci, co, P(x), P(dy), P(dw), P(bias), 0, P(ws), ws.numel() * 4), reps=5)
        mw, _ = timeit(lambda: lib.vc_conv3x3_wino_wgrad_f32(st(), B, H, H, ci, co, P(x), P(dy), P(dw), P(bias), 0, P(ws), ws.numel() * 4), reps=5)
        tot[0] += mb * mult
        tot[1] += mw * mult
        print("conv%s wgrad B=%d H=%3d %3d->%3d: split-bf16 direct %8.3f ms %6.1f TFLOP/s | f32 Winograd %8.3f ms %6.1f TFLOP/s" % (name, B, H, ci, co, mb, fl / mb, mw, fl / mw), flush=True)
    print("sum over the twelve layers B=%d wgrad: split-bf16 direct %8.3f ms | f32 Winograd %8.3f ms" % (B, tot[0], tot[1]))


def lstm():
    """marginal cost of one recurrence step = (sequence of T=42) - (sequence of T=2), / 40: the input projection and the
    weight-gradient GEMMs of the sequence drivers scale with T too, so they are timed separately with mode 1 and N fixed ... no:
    they are linear in T as well; the recurrent step kernels are isolated by timing the SAME driver in two modes and reporting both."""
    import os
    modes = [int(m) for m in os.environ.get("VC_LSTM_MODES", "3,1").split(",")]
    names = {3: "register-operand recurrence kernels", 2: "auto", 1: "gemm + gate kernels", 0: "round-1 fused step kernels"}
    H, E = 512, 256
    for N in [int(v) for v in os.environ.get("VC_LSTM_NS", "160,320,640,1280").split(",")]:
        res = {}
        for mode in modes:
            fl = (mode + 1) | (0x10 if os.environ.get("VC_PRECISION") == "bf16x3" else 0)   # VC_LSTM_KERNELS(mode) | VC_LSTM_BF16X3
            t = {}
            for T in (2, 42):
                X, W, b = rnd(T, N, E), rnd(E + H, 4 * H) * 0.05, rnd(4 * H) * 0.1
                lens = torch.full((N,), T, dtype=torch.int32, device="cuda")
                act, cs, hs = torch.empty(T, N, 4 * H, device="cuda"), torch.zeros(T + 1, N, H, device="cuda"), torch.zeros(T + 1, N, H, device="cuda")
                ws = torch.empty(lib.vc_lstm_seq_workspace_bytes(T, N, E, H) // 4 + 4, device="cuda")
                f, _ = timeit(lambda: lib.vc_lstm_seq_fwd_f32(st(), T, N, E, H, P(X), P(W), P(b), P(lens), P(act), P(cs), P(hs), P(ws), ws.numel() * 4, fl), reps=5)
                dhs = rnd(T + 1, N, H) * 0.1
                dH, dC, dG = torch.zeros(N, H, device="cuda"), torch.zeros(N, H, device="cuda"), torch.empty(T, N, 4 * H, device="cuda")
                dX, dW, db = torch.empty(T, N, E, device="cuda"), torch.empty(E + H, 4 * H, device="cuda"), torch.empty(4 * H, device="cuda")
                bw, _ = timeit(lambda: lib.vc_lstm_seq_bwd_f32(st(), T, N, E, H, P(X), P(W), P(lens), P(act), P(cs), P(hs), P(dhs), P(dH), P(dC), P(dG), P(dX), P(dW), P(db), P(ws), ws.numel() * 4, fl), reps=5)
                t[T] = (f, bw)
            res[mode] = ((t[42][0] - t[2][0]) / 40 * 1e3, (t[42][1] - t[2][1]) / 40 * 1e3, t[42][0], t[42][1])
            print("lstm N=%4d mode %d (%s): T=42 sequence fwd %.3f ms bwd %.3f ms; marginal per step (incl. its share of the T-linear GEMMs) fwd %.1f us bwd %.1f us"
                  % (N, mode, names[mode], res[mode][2], res[mode][3], res[mode][0], res[mode][1]), flush=True)


def mid():
    """129..383-tile shapes (VC_GEMM_FORCE_S sweeps the split count of the 128 x 128 path)"""
    for (ta, tb, M, N, K) in [(0, 1, 6400, 512, 10000), (1, 0, 512, 10000, 6400), (0, 1, 12800, 512, 10000), (0, 0, 12800, 512, 2048), (0, 0, 6400, 1024, 2048), (0, 0, 5120, 2048, 512), (1, 0, 2048, 2048, 28160)]:
        A = rnd(K, M) if ta else rnd(M, K)
        B = rnd(N, K) if tb else rnd(K, N)
        C = torch.empty(M, N, device="cuda")
        ws = torch.empty(max(lib.vc_gemm_workspace_bytes(M, N, K), 16) // 4 + 4, device="cuda")
        med, mn = timeit(lambda: lib.vc_gemm_f32(st(), ta, tb, M, N, K, P(A), M if ta else K, P(B), K if tb else N, P(C), N, None, 0, P(ws), ws.numel() * 4), reps=5)
        print("gemm ta=%d tb=%d %6d x %6d x %6d: %8.3f ms  %6.1f TFLOP/s  ws %.0f MB" % (ta, tb, M, N, K, med, 2e-9 * M * N * K / med, ws.numel() * 4 / 1e6))


def fc():
    """the six VGG fc GEMMs of a 64-image step (weight-bandwidth bound: fc1 = 411 MB, fc2 = 67 MB) + caption-side skinny shapes"""
    Bn = 64
    shapes = [("fc1 fwd", 0, 0, Bn, 4096, 25088), ("fc1 dgrad", 0, 1, Bn, 25088, 4096), ("fc1 wgrad", 1, 0, 25088, 4096, Bn),
              ("fc2 fwd", 0, 0, Bn, 4096, 4096), ("fc2 dgrad", 0, 1, Bn, 4096, 4096), ("fc2 wgrad", 1, 0, 4096, 4096, Bn),
              ("imf_emb fwd", 0, 0, Bn, 256, 4096), ("imf_emb wgrad", 1, 0, 4096, 256, Bn), ("z_rnn fwd", 0, 0, 320, 256, 15000),
              ("z_rnn wgrad", 1, 0, 15000, 256, 320), ("z_rnn dgrad", 0, 1, 320, 15000, 256)]
    for name, ta, tb, M, N, K in shapes:
        A = rnd(K, M) if ta else rnd(M, K)
        B = rnd(N, K) if tb else rnd(K, N)
        C = torch.empty(M, N, device="cuda")
        ws = torch.empty(max(lib.vc_gemm_workspace_bytes(M, N, K), 16) // 4 + 4, device="cuda")
        med, mn = timeit(lambda: lib.vc_gemm_f32(st(), ta, tb, M, N, K, P(A), M if ta else K, P(B), K if tb else N, P(C), N, None, 0, P(ws), ws.numel() * 4), reps=5)
        byts = 4.0 * (M * K + K * N + M * N)
        print("%-14s ta=%d tb=%d %6d x %6d x %6d: %8.3f ms  %6.1f TFLOP/s  %6.2f TB/s (operands once)  ws %.1f MB" % (
            name, ta, tb, M, N, K, med, 2e-9 * M * N * K / med, byts / med / 1e9, ws.numel() * 4 / 1e6))


if __name__ == "__main__":
    for a in sys.argv[1:] or ["gemm", "ablate", "conv", "lstm"]:
        globals()[a]()
