"""Round 6: where does a faster conv4_x / conv5_x kernel go?  Device time of the VGG16 forward and of the VGG16 backward alone (HIP events
on the caller's stream; both phases join their side streams before they return), 64 images, three streams, for VC_WINO4V=0 / 1 (set in
the environment: the switch is read per process).  python tools/experiments/vgg_phase_times.py [images]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_captioning_amd import abi, spec, synth  # noqa: E402
from vae_captioning_amd.trainer import Trainer  # noqa: E402
from vae_captioning_amd.utils.parameters import Parameters  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = Parameters()
p.fine_tune, p.batch_size = True, B
V, T = 10000, 20
rng = np.random.default_rng(0)
tr = Trainer(p, V, lib=abi.load(), precision=os.environ.get("VC_PRECISION", "f32"))
tr.load_state_dict({**spec.init_caption_params(p, V, seed=1), **spec.init_vgg_params(seed=2)})
tr.set_batch(synth.make_batch(rng, B, p.num_captions, T, V, images=True))
for _ in range(3):
    tr.train_step()
torch.cuda.synchronize()
vgg = tr.vgg
d = torch.randn(B, 4096, device="cuda") * 1e-3
tf, tb = [], []
for it in range(12):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    if os.environ.get("VC_FWD_ONE_CHAIN") == "1":   # (experiment: the forward as ONE chain of full-batch launches, the backward on three streams)
        keep, vgg._side = vgg._side, None
        vgg.forward(tr.images, tr.cap.step)
        vgg._side = keep
    else:
        vgg.forward(tr.images, tr.cap.step)
    e[1].record()
    vgg.backward(d)
    e[2].record()
    torch.cuda.synchronize()
    if it >= 2:
        tf.append(e[0].elapsed_time(e[1])); tb.append(e[1].elapsed_time(e[2]))
print("VC_WINO4V=%s precision=%s images=%d: VGG16 forward %.3f ms (min %.3f)  backward %.3f ms (min %.3f)" % (
    os.environ.get("VC_WINO4V", "default"), tr.precision, B, np.mean(tf), min(tf), np.mean(tb), min(tb)))
