import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vae_captioning_amd import abi, spec, synth
from vae_captioning_amd.engine import CaptionEngine
from vae_captioning_amd.generate import CaptionGenerator
import vae_captioning_amd.generate as G
w = dict(bench.WORKLOADS["cfg5"]); p = bench.make_params(w); p.vocab_size = bench.VOCAB
lib = abi.load()
eng = CaptionEngine(p, bench.VOCAB, lib=lib, seed=0); eng.load_params(spec.init_caption_params(p, bench.VOCAB, seed=1))
gen = CaptionGenerator(eng)
rng = np.random.default_rng(0); B = w["B"]
feats = torch.from_numpy(np.maximum(rng.standard_normal((B, 4096), dtype=np.float32), 0)).cuda()
cv = np.zeros((B, 90), np.float32); eps = rng.standard_normal((p.gen_z_samples, B, p.latent_size), dtype=np.float32)
run = lambda: gen.beam_search(feats, cv, eps, synth.BOS, synth.EOS, beam_size=w["beam"], max_len=p.gen_max_len)
for _ in range(4): run()
# instrument: time init_state and total
orig = gen.init_state
acc = {"init": 0.0, "n": 0}
def timed_init(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = orig(*a, **k); torch.cuda.synchronize(); acc["init"] += time.perf_counter() - t; acc["n"] += 1; return r
gen.init_state = timed_init
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / 20
print("per call %.3f ms; init_state %.3f ms (synchronised)" % (1e3 * tot, 1e3 * acc["init"] / acc["n"]))
G.PHASE_TIMES = {}
for _ in range(20): run()
print("phases (ms per call, synchronised at each boundary):", {k: round(1e3 * v / 20, 3) for k, v in G.PHASE_TIMES.items() if k})
G.PHASE_TIMES = None
import cProfile, pstats
gen.init_state = orig
pr = cProfile.Profile(); pr.enable()
for _ in range(10): run()
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
