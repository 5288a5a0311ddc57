"""-m gpu: the engine's data-parallel branches with TWO REAL PROCESSES.

tests/test_gpu_fullsize.py checks the world = 2 code path with both ranks as threads of one process (FakeGroup); here each rank is its
own process under torch.distributed.run, driving Trainer.train_step on its shard (tests/dp_worker.py) -- the label-count /
[mean | std] all-gather / [dmean | dstd] reduce-scatter / flat-gradient all-reduce sequence (four asynchronous buckets under the VGG16
backward when fine-tuning) crosses real process boundaries, in whatever order two unsynchronised processes reach it.  Both ranks
share the box's single GPU, so the process group is gloo (RCCL refuses two ranks on one device: RCCL with N > 1 stays unexecuted).
Nothing in the reference to mirror (utils/parameters.py:163-164: one GPU); the semantics are those of SURVEY.md section 8e.

Checks per case, after two optimiser steps:
  * the two replicas end bit-identical;
  * rank 0 equals, BIT FOR BIT, the in-process two-thread emulation of the same step (same shard arithmetic, same a + b sums:
    inter-process ordering may not change a single bit -- and the bucketed asynchronous all-reduce equals the single blocking one);
  * rank 0 agrees with ONE process training on the whole global batch (q1 'global': the reference's own semantics on the
    concatenated batch): per-step losses to 2e-4 relative, parameters to 2.5 learning rates per step (Adam's first steps move
    every weight by ~lr * sign(g): a gradient that is zero to rounding may take either sign)."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

from . import dp_worker
from .test_gpu_fullsize import FakeGroup

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 2


def _spawn(case, q1_mode, outdir, extra_env=None):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker.py"), case, q1_mode, str(STEPS), str(outdir)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return [dict(np.load(os.path.join(outdir, "rank%d.npz" % k))) for k in range(2)]


def _threads(lib, case, q1_mode):
    """The same two ranks as threads of THIS process (FakeGroup: in-process sums, the single blocking gradient all-reduce)."""
    from vae_captioning_amd import dp
    from vae_captioning_amd.trainer import Trainer
    p, V, P0, batch, noise, B = dp_worker.problem(case)
    fg = FakeGroup(2)
    res, errs = [None, None], []

    def run(r):
        try:
            torch.cuda.set_device(0)
            tr = Trainer(p, V, lib=lib, world=2, rank=r, seed=3)
            tr.cap.q1_mode = q1_mode
            tr.cap.reduce_fn, tr.cap.gather_fn, tr.cap.rscatter_fn = fg.hooks(r)
            tr.cap._fake_collectives = True
            tr.load_state_dict(P0)
            losses = []
            for _ in range(STEPS):
                tr.set_batch(dp.shard_batch(batch, r, 2, p.num_captions), dp_worker.shard_noise_all(noise, r, 2, B, p.num_captions, q1_mode))
                tr.train_step()
                losses.append(tr.losses())
            torch.cuda.synchronize()
            res[r] = dp_worker.result_of(tr, losses)
        except Exception as ex:  # pragma: no cover
            errs.append(ex)
            fg.bar.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    return res


def _single(lib, case):
    from vae_captioning_amd.trainer import Trainer
    p, V, P0, batch, noise, B = dp_worker.problem(case)
    tr = Trainer(p, V, lib=lib, seed=3)
    tr.load_state_dict(P0)
    losses = []
    for _ in range(STEPS):
        tr.set_batch(batch, noise)
        tr.train_step()
        losses.append(tr.losses())
    torch.cuda.synchronize()
    return dp_worker.result_of(tr, losses), p


@pytest.mark.parametrize("case,q1_mode", [("normal", "global"), ("ag", "global"), ("normal", "tower"), ("fine_tune", "global")])
def test_two_processes_train_like_one(lib, tmp_path, case, q1_mode):
    ranks = _spawn(case, q1_mode, tmp_path)
    assert set(ranks[0]) == set(ranks[1])
    for k in ranks[0]:
        if k != "#losses":   # (the reported losses are global values on every rank too, but only the parameters must be replicas)
            np.testing.assert_array_equal(ranks[0][k], ranks[1][k], err_msg="replicas differ: " + k)
    np.testing.assert_array_equal(ranks[0]["#losses"], ranks[1]["#losses"])
    emu = _threads(lib, case, q1_mode)
    for k in ranks[0]:
        np.testing.assert_array_equal(ranks[0][k], emu[0][k], err_msg="two processes != two threads: " + k)
    if q1_mode != "global":
        return   # the tower mix is a different (per-shard) z mix than a single process computes: nothing to compare with
    one, p = _single(lib, case)
    l2, l1 = ranks[0]["#losses"], one["#losses"]   # rows: steps; columns kld, rec_loss, lower_bound, annealing
    np.testing.assert_allclose(l2[:, :3], l1[:, :3], rtol=2e-4, atol=1e-6)
    for k in ranks[0]:
        if k.startswith("#") or k.endswith("#sum"):
            continue
        lr = p.cnn_lr if k.startswith("cnn/") else p.learning_rate
        d = np.abs(ranks[0][k].astype(np.float64) - one[k]).max()
        assert d <= 2.5 * lr * STEPS, (k, d)


def test_two_processes_in_the_split_bf16_mode(lib, tmp_path, monkeypatch):
    """the fine-tune case once more with VC_PRECISION=bf16x3 in both ranks: dense products, recurrences and the VGG16 weight gradients
    (csrc/conv_wgrad_bx.hip, feeding the four asynchronous gradient buckets from the weight-gradient stream) on the bf16 pipe -- the
    replicas stay bit-identical, two processes equal two threads bit for bit, and the result stays within Adam's sign noise of ONE
    process on the global batch in the same mode"""
    monkeypatch.setenv("VC_PRECISION", "bf16x3")   # (the default precision of every Trainer built below; per trainer, no library state)
    ranks = _spawn("fine_tune", "global", tmp_path, {"VC_PRECISION": "bf16x3"})
    for k in ranks[0]:
        np.testing.assert_array_equal(ranks[0][k], ranks[1][k], err_msg="replicas differ: " + k)
    emu = _threads(lib, "fine_tune", "global")
    assert lib.vc_gemm_get_precision() == 0   # nobody touched the deprecated process-wide default
    for k in ranks[0]:
        np.testing.assert_array_equal(ranks[0][k], emu[0][k], err_msg="two processes != two threads: " + k)
    one, p = _single(lib, "fine_tune")
    np.testing.assert_allclose(ranks[0]["#losses"][:, :3], one["#losses"][:, :3], rtol=5e-4, atol=1e-6)
    for k in ranks[0]:
        if k.startswith("#") or k.endswith("#sum"):
            continue
        lr = p.cnn_lr if k.startswith("cnn/") else p.learning_rate
        d = np.abs(ranks[0][k].astype(np.float64) - one[k]).max()
        assert d <= 2.5 * lr * STEPS, (k, d)
