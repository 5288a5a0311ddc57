import sys, torch
sys.path.insert(0, ".")
from vae_captioning_amd import abi
from vae_captioning_amd.abi import ptr as P
lib = abi.load()
st = lambda: torch.cuda.current_stream().cuda_stream
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N, K = 10000, 512
for M in (4096, 5120, 6016, 6144, 6272, 6400, 6528, 6656, 7680, 12800, 25600):
    A, B, C = torch.rand(M, K, device="cuda"), torch.rand(K, N, device="cuda"), torch.empty(M, N, device="cuda")
    ws = torch.empty(max(lib.vc_gemm_workspace_bytes(M, N, K), 16) // 4 + 4, device="cuda")
    ms = t(lambda: lib.vc_gemm_f32(st(), 0, 0, M, N, K, P(A), K, P(B), N, P(C), N, None, 0, P(ws), ws.numel() * 4))
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print("M=%5d tiles %5d  %.3f ms  %.1f TFLOP/s  us/tile %.3f" % (M, tiles, ms, 2e-9 * M * N * K / ms, 1e3 * ms / tiles))
