"""Experiment: does RCCL accept TWO ranks on ONE GPU (so that dp.AbiComm's multi-rank path can be exercised on a 1-GPU box)?
   python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/experiments/abi_comm_2rank_1gpu.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_captioning_amd import abi, dp  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = abi.load()
try:
    msg = [dp.AbiComm.unique_id(lib) if rank == 0 else None]
    dist.broadcast_object_list(msg, src=0)
    comm = dp.AbiComm(lib, world, rank, 0, msg[0])
    t = torch.full((1000,), float(rank + 1), device="cuda")
    comm.all_reduce(t)
    g = torch.empty(2 * 8, device="cuda")
    comm.all_gather(g, torch.full((8,), float(rank), device="cuda"))
    o = torch.empty(4, device="cuda")
    comm.reduce_scatter(o, torch.arange(8, device="cuda", dtype=torch.float32) + rank)
    torch.cuda.synchronize()
    print("rank", rank, "allreduce", float(t[0]), "gather", g.tolist()[::8], "rscatter", o.tolist(), "rccl", comm.rccl_version, flush=True)
    comm.destroy()
except Exception as e:  # noqa: BLE001
    print("rank", rank, "FAILED:", repr(e)[:500], flush=True)
dist.destroy_process_group()
