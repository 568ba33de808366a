#!/usr/bin/env python3
"""Can RCCL run the bench's one collective (gather of device tensors to rank 0) with two ranks on ONE GPU? Probe only."""
import os
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    x = torch.full((4, 8, 4), float(rank + 1), device="cuda:0")
    g = torch.empty((world, 4, 8, 4), device="cuda:0") if rank == 0 else None
    dist.gather(x, list(g.unbind(0)) if rank == 0 else None, dst=0)
    torch.cuda.synchronize()
    if rank == 0:
        print("gather ok", [float(g[i].mean()) for i in range(world)], flush=True)
    dist.barrier()
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}: {type(e).__name__}: {str(e)[:300]}", flush=True)
