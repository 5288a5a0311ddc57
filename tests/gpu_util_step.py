"""A small fine-tune Trainer for the subprocess tests of tests/test_gpu_orders.py (two images at 224x224, two captions each)."""
import numpy as np

from vae_captioning_amd import spec, synth
from vae_captioning_amd.trainer import Trainer
from vae_captioning_amd.utils.parameters import Parameters


def tiny_finetune_trainer(images=2):
    p = Parameters()
    p.fine_tune, p.num_captions, p.gen_z_samples = True, 2, 4
    V, B, T = 300, images, 6
    rng = np.random.default_rng(11)
    batch = synth.make_batch(rng, B, p.num_captions, T, V, images=True, variable_len=True)
    tr = Trainer(p, V, seed=3)
    tr.load_state_dict({**spec.init_caption_params(p, V, seed=1), **spec.init_vgg_params(seed=3)})
    tr.set_batch(batch)
    return tr
