import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from vae_captioning_amd import abi, spec, synth
from vae_captioning_amd.trainer import Trainer
from vae_captioning_amd.utils.parameters import Parameters
lib = abi.load()
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = Parameters(); p.fine_tune, p.batch_size = True, Bn
V, T = 10000, 20
rng = np.random.default_rng(4)
batch = synth.make_batch(rng, Bn, p.num_captions, T, V, images=True)
tr = Trainer(p, V, lib=lib, seed=3)
PV = spec.init_vgg_params(seed=2)
sc = np.float32(float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
PV = {k: (v * sc if "weights" in k else v) for k, v in PV.items()}
tr.load_state_dict({**spec.init_caption_params(p, V, seed=1), **PV})
tr.set_batch(batch)
cap, vgg = tr.cap, tr.vgg
def forward(train):
    cap.step.zero_()
    feats = vgg.forward(tr.images, cap.step)
    vgg.reg_sumsq(cap.red.data_ptr() + 12)
    cap.forward(feats, train=train)
forward(True)
dfe = cap.backward(want_dfeatures=True); vgg.backward(dfe); torch.cuda.synchronize()
print("losses train", cap.out.tolist())
forward(False); print("losses eval ", cap.out.tolist())
g = vgg.store.g[:vgg.store.n].clone(); p0 = vgg.store.p.clone()
print("fc2 abs max", float(vgg.buf["fc2"].abs().max()), "mean", float(vgg.buf["fc2"].mean()), "imf abs max", float(cap.buf["imf"].abs().max()))
for name in ["cnn/fc1/weights", "cnn/conv3_1/weights", "ALL"]:
    if name == "ALL":
        d = g.clone()
    else:
        off, shape = vgg.store.offsets[name]; n = int(np.prod(shape))
        d = torch.zeros_like(g); d[off:off + n] = g[off:off + n]
    gn2 = float((d.double() ** 2).sum())
    for scale in (1e-2, 4e-3, 2e-3):
        eps = scale / np.sqrt(gn2)
        vals = []
        for sgn in (1, -1):
            vgg.store.p.copy_(p0 + sgn * eps * d); forward(False); vals.append(float(cap.out[2].item()))
        vgg.store.p.copy_(p0)
        print("%-28s |g|^2 %.4e  eps*|g| %.0e  fd %.4e  ratio %.3f" % (name, gn2, scale, (vals[0] - vals[1]) / (2 * eps), (vals[0] - vals[1]) / (2 * eps) / gn2))
