#!/usr/bin/env python
"""Feasibility probe for the non-fused Winograd F(4x4,3x3) route (round 6): the 36 per-position products M_p[n][t] = U_p[n][c] V_p[c][t]
of a layer as ONE product of the same tile count (N = 36 x tiles), f32 MFMA against split-bf16, beside the fused kernel's time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.microbench import lib, P, st, timeit, rnd  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, hw, ci, co in (("conv3_2", 56, 256, 256), ("conv4_1", 28, 256, 512), ("conv4_2", 28, 512, 512), ("conv5_2", 14, 512, 512)):
    tiles = nb * ((hw + 3) // 4) ** 2
    M, N, K = co, 36 * tiles, ci
    A, B, C = rnd(M, K), rnd(K, N), torch.empty(M, N, device="cuda")
    ws = torch.empty(max(lib.vc_gemm_workspace_bytes(M, N, K), 16) // 4 + 4, device="cuda")
    res = []
    for fn in (lib.vc_gemm_f32, lib.vc_gemm_bf16x3_f32):
        med, mn = timeit(lambda: fn(st(), 0, 0, M, N, K, P(A), K, P(B), N, P(C), N, None, 0, P(ws), ws.numel() * 4))
        res.append(med)
    # transposed orientation: tiles as rows
    A2, B2, C2 = rnd(N, K), rnd(K, M), torch.empty(N, M, device="cuda")
    ws2 = torch.empty(max(lib.vc_gemm_workspace_bytes(N, M, K), 16) // 4 + 4, device="cuda")
    for fn in (lib.vc_gemm_f32, lib.vc_gemm_bf16x3_f32):
        med, mn = timeit(lambda: fn(st(), 0, 0, N, M, K, P(A2), K, P(B2), M, P(C2), M, None, 0, P(ws2), ws2.numel() * 4))
        res.append(med)
    fl = 2e-9 * M * N * K
    vbytes = 4.0 * 36 * tiles * (ci + co)
    print("%s %d img: tiles %d  GEMM [%d x %d x %d]: f32 %.3f ms (%.0f TF)  bf16x3 %.3f ms | tiles-as-rows f32 %.3f  bf16x3 %.3f | V+M %.0f MB -> x2 passes at 4 TB/s %.3f ms"
          % (name, nb, tiles, M, N, K, res[0], fl / res[0], res[1], res[2], res[3], vbytes / 1e6, 2 * vbytes / 4e9), flush=True)
