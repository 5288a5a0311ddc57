"""-m gpu, TWO OR MORE GPUs: the data-parallel step over RCCL with one GPU per rank.  SKIPPED on the build lease (one GPU per box):
RCCL with more than one rank has never executed in this project and no scaling curve has been measured -- these tests are what runs
the moment a multi-GPU box does (DESIGN.md section 5 holds the exposed-communication budget the first curve is to be read against).

  * two processes under torch.distributed.run with backend nccl (= RCCL), one GPU each, collectives through libvaecap's own
    communicator (vc_comm_init_rank / vc_allreduce_sum_f32 / vc_allgather_f32 / vc_reducescatter_sum_f32; tests/dp_worker.py asserts
    vc_comm_info reports world 2): replicas bit-identical, bit-identical to the in-process two-thread emulation (a two-rank sum is one
    addition: RCCL's order cannot differ), equal to one process on the global batch within Adam's sign noise -- the properties
    tests/test_gpu_dp_procs.py proves over gloo on one GPU;
  * `python bench.py --gpus 2` launches its ranks, and rank 0's line reports an RCCL world of two through the C ABI.
The reference is single-GPU (utils/parameters.py:163-164); semantics are SURVEY.md section 8e's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (one per RCCL rank); the build lease has one")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case,precision", [("normal", "f32"), ("ag", "f32"), ("fine_tune", "f32"), ("fine_tune", "bf16x3")])
def test_two_ranks_over_rccl_train_like_one(lib, tmp_path, monkeypatch, case, precision):
    from .test_gpu_dp_procs import STEPS, _single, _spawn, _threads
    env = {"VC_DP_BACKEND": "nccl", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    if precision != "f32":
        env["VC_PRECISION"] = precision
        monkeypatch.setenv("VC_PRECISION", precision)
    ranks = _spawn(case, "global", tmp_path, env)
    for k in ranks[0]:
        np.testing.assert_array_equal(ranks[0][k], ranks[1][k], err_msg="replicas differ: " + k)
    emu = _threads(lib, case, "global")
    for k in ranks[0]:
        np.testing.assert_array_equal(ranks[0][k], emu[0][k], err_msg="two RCCL ranks != two threads: " + k)
    one, p = _single(lib, case)
    np.testing.assert_allclose(ranks[0]["#losses"][:, :3], one["#losses"][:, :3], rtol=5e-4, atol=1e-6)
    for k in ranks[0]:
        if k.startswith("#") or k.endswith("#sum"):
            continue
        lr = p.cnn_lr if k.startswith("cnn/") else p.learning_rate
        assert np.abs(ranks[0][k].astype(np.float64) - one[k]).max() <= 2.5 * lr * STEPS, k


@pytest.mark.parametrize("workload", ["cfg2", "cfg4"])
def test_bench_on_two_gpus_reports_an_rccl_world_of_two(workload):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--workload", workload,
                        "--no-cpu-baseline", "--strong-n1", "0"], env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["rccl_world_size"] == 2 and line["config"]["parallelism"] == "dp2"
    dp = line["dp"] if "dp" in line else line.get("data_parallel", {})
    assert dp.get("backend") == "nccl" and dp.get("rccl_world_size") == 2 and "libvaecap C ABI" in dp.get("collectives", ""), dp
    assert dp.get("exposed_comm_ms") is not None and len(dp.get("buckets_bytes", [])) == (4 if workload == "cfg4" else 1)
