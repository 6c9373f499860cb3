#!/usr/bin/env python3
"""One-rank RCCL probe of the collective the 2-D layout uses on a multi-GPU node: all_to_all_single on the DEFAULT communicator
with explicit split sizes (cellmapper_amd/sharded.py, WorldSlice) -- run under torch.distributed.run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
dev = torch.device("cuda", torch.cuda.current_device())
dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
n = 1 << 20
a = torch.arange(n * world, dtype=torch.int64, device=dev) + rank * 1000003
out = torch.empty_like(a)
dist.all_to_all_single(out, a, output_split_sizes=[n] * world, input_split_sizes=[n] * world)
torch.cuda.synchronize()
ok = bool((out[rank * n:(rank + 1) * n] == a[rank * n:(rank + 1) * n]).all())
print(f"[rccl probe] rank {rank}/{world}: all_to_all_single with split sizes on the default communicator: {'ok' if ok else 'MISMATCH'}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
