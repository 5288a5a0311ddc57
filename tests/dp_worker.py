"""One RANK of the two-process data-parallel parity test (tests/test_gpu_dp_procs.py starts two of these under
torch.distributed.run; both share the test box's single GPU, so the process group is gloo -- RCCL refuses two ranks on one device).
Each rank drives Trainer.train_step on ITS shard of a seeded global batch; the collectives (count / [mean|std] all-gather / [dmean|dstd]
reduce-scatter / the flat-gradient all-reduce, as four asynchronous buckets when fine-tuning) cross REAL process boundaries.
Every rank writes its result; the test compares them with in-process runs.  Not a pytest file."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def problem(case):
    """-> (p, V, P0, batch, noise, B): the seeded GLOBAL problem every process rebuilds identically."""
    from vae_captioning_amd import spec, synth
    from vae_captioning_amd.utils.parameters import Parameters
    p = Parameters()
    p.embed_size, p.encoder_hidden, p.decoder_hidden = 32, 64, 64
    p.latent_size, p.gen_z_samples, p.cnn_feature_size = 12, 5, (4096 if case == "fine_tune" else 40)
    p.num_captions, p.batch_size = 3, 4
    p.prior, p.use_c_v = ("AG", True) if case == "ag" else ("Normal", False)
    p.fine_tune = case == "fine_tune"
    p.lstm_clip_by_norm = 0.05
    V, B, T = 150, 4, 6
    rng = np.random.default_rng(11)
    P0 = spec.init_caption_params(p, V, seed=5)
    if p.fine_tune:
        P0.update({k: (v * np.float32(0.7) if "weights" in k else v) for k, v in spec.init_vgg_params(seed=2).items()})
    batch = synth.make_batch(rng, B, p.num_captions, T, V, use_ci=spec.uses_ci(p), variable_len=True, feature_size=p.cnn_feature_size,
                             images=p.fine_tune)
    noise = synth.make_noise(rng, B * p.num_captions, T, p)
    if p.fine_tune:   # injected fc dropout masks, so that every process drops the same units of the same image
        noise["cnn_drop1"] = (rng.random((B, 4096)) < 0.5).astype(np.float32)
        noise["cnn_drop2"] = (rng.random((B, 4096)) < 0.5).astype(np.float32)
    return p, V, P0, batch, noise, B


KEEP_VGG = ("cnn/conv1_1/weights", "cnn/conv1_1/biases", "cnn/conv3_1/weights", "cnn/conv5_3/biases_conv", "cnn/fc2/biases")


def result_of(tr, steps_losses):
    sd = tr.state_dict()
    out = {k: v for k, v in sd.items() if not k.startswith("cnn/") or k in KEEP_VGG}
    if "cnn/fc1/weights" in sd:  # 411 MB: keep a checksum and a slice
        out["cnn/fc1/weights#rows0-3"] = sd["cnn/fc1/weights"][:4].copy()
        out["cnn/fc1/weights#sum"] = np.array([sd["cnn/fc1/weights"].astype(np.float64).sum()])
    out["#losses"] = np.array(steps_losses, np.float64)
    return out


def shard_noise_all(noise, rank, world, B, nc, q1_mode):
    from vae_captioning_amd import dp
    sh = dp.shard_noise({k: v for k, v in noise.items() if not k.startswith("cnn_drop")}, rank, world, B * nc, q1_mode)
    b0, b1 = rank * (B // world), (rank + 1) * (B // world)
    for k in ("cnn_drop1", "cnn_drop2"):
        if k in noise:
            sh[k] = noise[k][b0:b1]
    return sh


def main():
    case, q1_mode, steps, outdir = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    from vae_captioning_amd import abi, dp
    from vae_captioning_amd.trainer import Trainer
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    backend = os.environ.get("VC_DP_BACKEND", "gloo")
    # gloo: both ranks on GPU 0 (the build box has one); nccl (= RCCL): one GPU per rank, collectives through libvaecap's vc_comm_* entries
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    lib = abi.load()
    p, V, P0, batch, noise, B = problem(case)
    tr = Trainer(p, V, lib=lib, world=world, rank=rank, seed=3)
    if backend == "nccl":   # the collectives must be the C ABI's own RCCL communicator over `world` ranks, not a fallback
        import ctypes
        w, r, ver = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        assert tr.comm is not None, "RCCL-backed process group, but the trainer did not bring up libvaecap's communicator"
        lib.vc_comm_info(tr.comm.h, ctypes.byref(w), ctypes.byref(r), ctypes.byref(ver))
        assert (w.value, r.value) == (world, rank), (w.value, r.value)
    tr.cap.q1_mode = q1_mode
    tr.load_state_dict(P0)
    losses = []
    for s in range(steps):
        tr.set_batch(dp.shard_batch(batch, rank, world, p.num_captions), shard_noise_all(noise, rank, world, B, p.num_captions, q1_mode))
        tr.train_step()
        losses.append(tr.losses())
    torch.cuda.synchronize()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **result_of(tr, losses))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
