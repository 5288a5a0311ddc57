"""Per-layer memory-side traffic of the 3x3 convolution kernels (what `roofline.traffic` sums over the step), one launch per call:
the twelve Winograd layers of VGG16 at 64 images (B=... in the environment), in program order
  forward (F(4x4,3x3), bias + ReLU + mask bits, + the fused pool where VGG16 has one), data gradient (mask as bits), weight gradient (F(3x3,2x2)).
Workload:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR/p1 -- python tools/traffic_probe_layers.py
           rocprofv3 --pmc WRITE_SIZE ...                                 -d DIR/p2 -- (same)
Summary:   python tools/traffic_probe_layers.py --summary DIR   (tools/traffic_probe_layers.sh runs all three)
Counter units and the gfx950 correction as MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE in KiB; reads x 2: 16-byte-per-lane streams)."""
import csv
import glob
import os
import sys
from collections import defaultdict

LAYERS = [("conv1_2", 224, 64, 64, True), ("conv2_1", 112, 64, 128, False), ("conv2_2", 112, 128, 128, True), ("conv3_1", 56, 128, 256, False),
          ("conv3_2", 56, 256, 256, False), ("conv3_3", 56, 256, 256, True), ("conv4_1", 28, 256, 512, False), ("conv4_2", 28, 512, 512, False),
          ("conv4_3", 28, 512, 512, True), ("conv5_1", 14, 512, 512, False), ("conv5_2", 14, 512, 512, False), ("conv5_3", 14, 512, 512, True)]
B = int(os.environ.get("B", "64"))
ROUNDS = 2


def algorithmic(kind, H, ci, co, pool):
    """bytes a launch has to move: (read, written)"""
    px = B * H * H * 4
    if kind == "fwd":
        return px * ci + 36 * ci * co * 4, px * co + (px * co // 4 + px * co // 128 if pool else px * co // 32)
    if kind == "dgrad":
        return px * co + 36 * ci * co * 4 + px * ci // 32, px * ci
    return px * (ci + co), 9 * ci * co * 4


def summary(root):
    got = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        per = defaultdict(float)
        order = []
        for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != c:
                    continue
                n = r["Kernel_Name"]
                if not ("conv_wino4_kernel" in n or "wino_wgrad" in n):
                    continue
                d = int(r["Dispatch_Id"])
                if d not in per:
                    order.append((d, n))
                per[d] += float(r["Counter_Value"])
        got[c] = (per, sorted(set(order)))
    # program order of a round: 12 forward, 12 data gradients, 12 x (weight gradient main [+ reduce])
    rows = {}
    for c, (per, order) in got.items():
        w4 = [d for d, n in order if "conv_wino4_kernel" in n]
        wg = [(d, n) for d, n in order if "wino_wgrad" in n]
        nr = len(w4) // 24
        assert nr >= 1 and len(w4) == nr * 24, (c, len(w4))
        w4 = w4[-24:]                                  # the last round
        # weight gradient: group each main kernel with the reduce launches behind it
        groups = []
        for d, n in wg:
            if "reduce" in n:
                groups[-1].append(d)
            else:
                groups.append([d])
        groups = groups[-12:]
        for i, L in enumerate(LAYERS):
            rows.setdefault((L[0], "fwd"), {})[c] = per[w4[i]] * 1024
            rows.setdefault((L[0], "dgrad"), {})[c] = per[w4[12 + i]] * 1024
            rows.setdefault((L[0], "wgrad"), {})[c] = sum(per[d] for d in groups[i]) * 1024
    tot = defaultdict(lambda: [0.0, 0.0, 0.0, 0.0])
    print("| layer | call | read MB (2 x FETCH_SIZE) | algorithmic read MB | x | written MB | algorithmic written MB | x |")
    print("|---|---|---|---|---|---|---|---|")
    for kind in ("fwd", "dgrad", "wgrad"):
        for (name, H, ci, co, pool) in LAYERS:
            v = rows[(name, kind)]
            rd, wr = 2 * v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
            ar, aw = algorithmic(kind, H, ci, co, pool)
            t = tot[kind]
            t[0] += rd; t[1] += ar; t[2] += wr; t[3] += aw
            print("| %s | %s | %.1f | %.1f | %.2f | %.1f | %.1f | %.2f |" % (name, kind, rd / 1e6, ar / 1e6, rd / ar, wr / 1e6, aw / 1e6, wr / aw))
    for kind, t in tot.items():
        print("| all twelve | %s | %.0f | %.0f | %.2f | %.0f | %.0f | %.2f |" % (kind, t[0] / 1e6, t[1] / 1e6, t[0] / t[1], t[2] / 1e6, t[3] / 1e6, t[2] / t[3]))
    a = [sum(t[i] for t in tot.values()) for i in range(4)]
    print("\nsum: %.2f GB moved (%.2f read + %.2f written) against %.2f GB algorithmic = %.2f x" % (
        (a[0] + a[2]) / 1e9, a[0] / 1e9, a[2] / 1e9, (a[1] + a[3]) / 1e9, (a[0] + a[2]) / (a[1] + a[3])))


def workload():
    import torch
    sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
    from vae_captioning_amd import abi
    from vae_captioning_amd.abi import ptr as P
    lib = abi.load()
    st = lambda: torch.cuda.current_stream().cuda_stream
    ws_bytes = max(lib.vc_conv3x3_wino_wgrad_workspace_bytes(B, H, H, ci, co) for _, H, ci, co, _ in LAYERS)
    ws = torch.empty(ws_bytes // 4 + 16, device="cuda")
    cases = []
    for name, H, ci, co, pool in LAYERS:
        x = torch.rand(B * H * H * ci, device="cuda") * 2 - 1
        w = torch.rand(3, 3, ci, co, device="cuda") * 2 - 1
        bias = torch.rand(co, device="cuda")
        y = torch.empty(B * H * H * co, device="cuda")
        dx = torch.empty(B * H * H * ci, device="cuda")
        pl = torch.empty(B * H * H * co // 4, device="cuda") if pool else None
        pb = torch.empty(lib.vc_conv3x3_wino_pool_words(B, H, H, co) + 16, dtype=torch.int32, device="cuda") if pool else None
        mk = torch.empty(lib.vc_conv3x3_wino4_mask_words(B, H, H, co), dtype=torch.int32, device="cuda")
        mi = torch.zeros(lib.vc_conv3x3_wino4_mask_words(B, H, H, ci), dtype=torch.int32, device="cuda")
        mi.random_(0, 2 ** 31 - 1)
        vp, vpt = torch.empty(36 * ci * co, device="cuda"), torch.empty(36 * ci * co, device="cuda")
        dw, db = torch.empty(9 * ci * co, device="cuda"), torch.empty(co, device="cuda")
        lib.vc_conv3x3_wino4_pack_f32(st(), ci, co, P(w), 0, P(vp))
        lib.vc_conv3x3_wino4_pack_f32(st(), ci, co, P(w), 1, P(vpt))
        cases.append((H, ci, co, x, bias, y, dx, pl, pb, mk, mi, vp, vpt, dw, db))
    for _ in range(ROUNDS):
        for (H, ci, co, x, bias, y, dx, pl, pb, mk, mi, vp, vpt, dw, db) in cases:
            if pl is not None:
                rc = lib.vc_conv3x3_wino4_fwd_pool_f32(st(), B, H, H, ci, co, P(x), P(vp), P(bias), P(y), P(pl), P(pb))
            else:
                rc = lib.vc_conv3x3_wino4_fwd_mask_f32(st(), B, H, H, ci, co, P(x), P(vp), P(bias), P(y), 1, P(mk))
            assert rc == 0, rc
        for (H, ci, co, x, bias, y, dx, pl, pb, mk, mi, vp, vpt, dw, db) in cases:
            rc = lib.vc_conv3x3_wino4_dgrad_bits_f32(st(), B, H, H, ci, co, P(y), P(vpt), P(mi), P(dx))
            assert rc == 0, rc
        for (H, ci, co, x, bias, y, dx, pl, pb, mk, mi, vp, vpt, dw, db) in cases:
            rc = lib.vc_conv3x3_wino_wgrad_f32(st(), B, H, H, ci, co, P(x), P(y), P(dw), P(db), 0, P(ws), ws_bytes)
            assert rc == 0, rc
        torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summary":
        summary(sys.argv[2])
    else:
        workload()
