"""Do HIP stream priorities protect a critical-path GEMM from an off-chain one launched next to it?  (MI355X, through torch streams.)
critical: douts = dlogits . W^T  [6400,10000]x[10000,512];  off-chain: dW = outs^T . dlogits  [512,6400]x[6400,10000]"""
import sys, torch
sys.path.insert(0, ".")
from vae_captioning_amd import abi
from vae_captioning_amd.abi import ptr as P
lib = abi.load()
R, H, V = 6400, 512, 10000
dl = torch.randn(R, V, device="cuda"); W = torch.randn(H, V, device="cuda"); outs = torch.randn(R, H, device="cuda")
dh = torch.empty(R, H, device="cuda"); dW = torch.empty(H, V, device="cuda")
ws1 = torch.empty(64 << 20, device="cuda"); ws2 = torch.empty(64 << 20, device="cuda")
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range (least, greatest):", lo, hi)


def crit(s):
    lib.vc_gemm_f32(s.cuda_stream, 0, 1, R, H, V, P(dl), V, P(W), V, P(dh), H, None, 0, P(ws1), ws1.numel() * 4)


def off(s):
    lib.vc_gemm_f32(s.cuda_stream, 1, 0, H, V, R, P(outs), H, P(dl), V, P(dW), V, None, 0, P(ws2), ws2.numel() * 4)


def run(pc, po, both=True, reps=20):
    sc, so = torch.cuda.Stream(priority=pc), torch.cuda.Stream(priority=po)
    tc = tw = 0.0
    for i in range(reps + 3):
        torch.cuda.synchronize()
        e0, e1, w0, w1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        w0.record(so)
        if both:
            off(so)   # the competitor is ALREADY queued when the critical GEMM arrives
        e0.record(sc); crit(sc); e1.record(sc)
        w1.record(so)
        torch.cuda.synchronize()
        if i >= 3:
            tc += e0.elapsed_time(e1); tw += w0.elapsed_time(w1)
    return tc / reps * 1e3, tw / reps * 1e3


print("critical alone                      : %.0f us" % run(0, 0, both=False)[0])
for pc, po in ((0, 0), (-1, 0), (0, -1), (hi, lo)):
    c, w = run(pc, po)
    print("critical prio %2d, off-chain prio %2d : critical %.0f us, pair %.0f us" % (pc, po, c, w))
