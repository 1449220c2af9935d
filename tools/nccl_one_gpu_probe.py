#!/usr/bin/env python3
"""Can two ranks of one process group share ONE MI355X over RCCL?  (VERDICT r4 item 8 asks for the `nccl` all_gather path of bench.py / star_amd/multi_gpu.py to run on hardware
once, with both ranks on device 0.)  Spawns two processes, both on cuda:0, backend nccl, one all_gather of 32-byte records; prints what happened."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def work(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    try:
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
        x = torch.full((4, 32), rank, dtype=torch.uint8, device="cuda:0")
        out = [torch.empty_like(x) for _ in range(2)]
        dist.all_gather(out, x)
        torch.cuda.synchronize()
        print("rank %d: all_gather over nccl with both ranks on cuda:0 OK: %s" % (rank, [int(o[0, 0]) for o in out]), flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print("rank %d: FAILED: %s" % (rank, repr(e)[:400]), flush=True)


if __name__ == "__main__":
    mp.spawn(work, args=(29571,), nprocs=2, join=True)
