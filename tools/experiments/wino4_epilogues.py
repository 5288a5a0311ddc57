import sys, torch
sys.path.insert(0, ".")
from vae_captioning_amd import abi
from vae_captioning_amd.abi import ptr as P
lib = abi.load()
st = lambda: torch.cuda.current_stream().cuda_stream
B = 64
for (H, ci, co) in ((224, 64, 64), (112, 128, 128)):
    x = torch.randn(B * H * H * ci, device="cuda"); w = torch.randn(3, 3, ci, co, device="cuda") * 0.05; bias = torch.randn(co, device="cuda")
    y = torch.empty(B * H * H * co, device="cuda"); pl = torch.empty(B * H * H * co // 4, device="cuda")
    pb = torch.empty(lib.vc_conv3x3_wino_pool_words(B, H, H, co) + 16, dtype=torch.int32, device="cuda")
    mk = torch.empty(lib.vc_conv3x3_wino4_mask_words(B, H, H, co), dtype=torch.int32, device="cuda")
    vp = torch.empty(36 * ci * co, device="cuda")
    lib.vc_conv3x3_wino4_pack_f32(st(), ci, co, P(w), 0, P(vp))
    fs = {"plain": lambda: lib.vc_conv3x3_wino4_fwd_f32(st(), B, H, H, ci, co, P(x), P(vp), P(bias), P(y), None, 1),
          "mask": lambda: lib.vc_conv3x3_wino4_fwd_mask_f32(st(), B, H, H, ci, co, P(x), P(vp), P(bias), P(y), 1, P(mk)),
          "pool(no codes)": lambda: lib.vc_conv3x3_wino4_fwd_f32(st(), B, H, H, ci, co, P(x), P(vp), P(bias), P(y), P(pl), 1),
          "pool+codes": lambda: lib.vc_conv3x3_wino4_fwd_pool_f32(st(), B, H, H, ci, co, P(x), P(vp), P(bias), P(y), P(pl), P(pb))}
    for k, f in fs.items():
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        print(H, ci, co, k, "%.3f ms" % (e0.elapsed_time(e1) / 10))
