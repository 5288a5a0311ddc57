"""Round 6: what would split-bf16 ("bf16x3") products cost in the Winograd F(4x4,3x3) DOMAIN?  (The review's non-fused route: transform ->
36 batched GEMMs on gemm_bx_kernel -> output transform.)  Emulation in torch on the GPU, conv4_2's shape: V = B^T d B and U = G g G^T in
f32, the per-position products as (i) f32, (ii) hi.hi + hi.lo + lo.hi of bf16 splits with f32 accumulation, the output transform in
f32; error of the layer's output against an fp64 direct convolution, relative to the tensor maximum."""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
dev = "cuda"
B, H, C, N = 4, 28, 512, 512
x = torch.relu(torch.randn(B, C, H, H, device=dev))
w = torch.randn(N, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
ref = F.conv2d(x.double(), w.double(), padding=1)
Bt = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float32, device=dev)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float32, device=dev)
At = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float32, device=dev)
xp = F.pad(x, (1, 1, 1, 1))
tiles = xp.unfold(2, 6, 4).unfold(3, 6, 4)                      # [B, C, 7, 7, 6, 6]
V = torch.einsum("ij,bcyxjk,lk->bcyxil", Bt, tiles, Bt)         # B^T d B
U = torch.einsum("ij,ncjk,lk->ncil", G, w, G)                   # G g G^T
Vm = V.permute(4, 5, 1, 0, 2, 3).reshape(36, C, -1)             # [p, c, t]
Um = U.permute(2, 3, 0, 1).reshape(36, N, C)                    # [p, n, c]


def split(a):
    hi = a.to(torch.bfloat16).float()
    lo = (a - hi).to(torch.bfloat16).float()
    return hi, lo


def finish(M):
    M = M.reshape(6, 6, N, B, 7, 7)
    Y = torch.einsum("ai,ijnbyx,cj->nbyxac", At, M, At)           # A^T m A -> [n, b, ty, tx, 4, 4]
    return Y.permute(1, 0, 2, 4, 3, 5).reshape(B, N, 28, 28)


torch.backends.cuda.matmul.allow_tf32 = False
M32 = torch.bmm(Um, Vm)
uh, ul = split(Um)
vh, vl = split(Vm)
Mbx = torch.bmm(uh, vh) + torch.bmm(uh, vl) + torch.bmm(ul, vh)
M64 = torch.bmm(Um.double(), Vm.double()).float()
mx = ref.abs().max().item()
for name, M in (("f32 products", M32), ("split-bf16 products (hi.hi + hi.lo + lo.hi)", Mbx), ("exact products of the f32 transforms", M64)):
    err = (finish(M).double() - ref).abs().max().item()
    print("%-48s max|err| %.3e = %.2e of the tensor maximum %.2f" % (name, err, err / mx, mx))
# the direct (non-Winograd) product in split-bf16, for scale: what the split costs without the transforms' amplification
xh, xl = split(x); wh, wl = split(w)
d = F.conv2d(xh, wh, padding=1) + F.conv2d(xh, wl, padding=1) + F.conv2d(xl, wh, padding=1)
print("%-48s max|err| %.3e = %.2e of the tensor maximum" % ("direct convolution, split-bf16 products", (d.double() - ref).abs().max().item(), (d.double() - ref).abs().max().item() / mx))
