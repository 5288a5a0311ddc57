#!/usr/bin/env python
"""Host-side enqueue time of one cfg4 training step (how close the Python / C-ABI launch path is to the 32 ms the GPU needs):
the step is enqueued N times WITHOUT synchronising in between while the device queue is kept short by a sync every step boundary
measurement: t_host = wall time of train_step() calls when the GPU is idle at the start of each call."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vae_captioning_amd import abi, spec, synth
from vae_captioning_amd.trainer import Trainer
lib = abi.load()
w = dict(bench.WORKLOADS["cfg4"]); p = bench.make_params(w)
tr = Trainer(p, 10000, lib=lib, seed=1)
tr.load_state_dict({**spec.init_caption_params(p, 10000, seed=1), **spec.init_vgg_params(seed=2)})
tr.set_batch(synth.make_batch(np.random.default_rng(0), 64, 5, 20, 10000, images=True))
for _ in range(3):
    tr.train_step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.train_step(); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("host enqueue time per cfg4 step: median %.2f ms, min %.2f ms (GPU idle at the start of every call)" % (1e3 * np.median(ts), 1e3 * min(ts)))
