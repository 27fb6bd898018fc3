#!/usr/bin/env python3
"""RCCL on this box: process group of one rank, barrier, all-reduce, and a batch_isend_irecv with no ops skipped.
(The multi-rank path of bench.py needs more than one GPU; this only shows that the backend initialises.)"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
t = torch.ones(4, device=dev)
dist.all_reduce(t)
dist.barrier()
torch.cuda.synchronize()
print("rccl ok:", t.tolist(), dist.get_backend())
dist.destroy_process_group()
