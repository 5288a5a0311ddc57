import sys, numpy as np, torch
sys.path.insert(0, '.')
from vae_captioning_amd import abi
from oracle import vgg as OV
from tests.gpu_util import P, dev, dev_c4, host_c4, stream, zeros
lib = abi.load()
B, H, W, Ci, Co = [int(v) for v in sys.argv[1:6]]
rng = np.random.default_rng(1)
x = np.maximum(rng.standard_normal((B, H, W, Ci), dtype=np.float32), 0)
w = (rng.standard_normal((3, 3, Ci, Co), dtype=np.float32) / np.sqrt(9 * Ci)).astype(np.float32)
ref = OV.conv3x3_fwd(x.astype(np.float64), w.astype(np.float64), np.zeros(Co))
wd = dev(w)
wp = torch.empty(lib.vc_conv3x3_bx_pack_bytes(Ci, Co) // 4, dtype=torch.float32, device="cuda")
lib.vc_conv3x3_bx_pack_f32(stream(), Ci, Co, P(wd), 0, P(wp))
y = zeros(B, Co // 4, H, W, 4)
lib.vc_conv3x3_bx_fwd_f32(stream(), B, H, W, Ci, Co, P(dev_c4(x)), P(wp), None, P(y), 0)
got = host_c4(y, (B, H, W, Co))
bad = np.abs(got - ref) > 1e-3 * np.abs(ref).max()
print("bad", bad.sum(), "of", bad.size)
idx = np.argwhere(bad)
for ax, nm in enumerate("byxc"):
    u, c = np.unique(idx[:, ax], return_counts=True)
    print(nm, dict(zip(u.tolist(), c.tolist())))
print(idx[:10], got[bad][:10], ref[bad][:10])
