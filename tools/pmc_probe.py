import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vae_captioning_amd import abi
from vae_captioning_amd.abi import ptr as P
lib = abi.load()
st = lambda: torch.cuda.current_stream().cuda_stream
B, H, ci, co = 64, 56, 256, 256
x = torch.rand(B, H, H, ci, device="cuda") * 2 - 1
w = torch.rand(3, 3, ci, co, device="cuda") * 2 - 1
bias = torch.rand(co, device="cuda")
y = torch.empty(B, H, H, co, device="cuda")
A = torch.rand(8192, 8192, device="cuda") * 2 - 1
Bm = torch.rand(8192, 8192, device="cuda") * 2 - 1
C = torch.empty(8192, 8192, device="cuda")
for _ in range(3):
    lib.vc_conv3x3_fwd_f32(st(), B, H, H, ci, co, P(x), P(w), P(bias), P(y), 1, None, 0)
    lib.vc_gemm_f32(st(), 0, 0, 8192, 8192, 8192, P(A), 8192, P(Bm), 8192, P(C), 8192, None, 0, None, 0)
torch.cuda.synchronize()
