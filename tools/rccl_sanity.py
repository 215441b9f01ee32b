#!/usr/bin/env python3
"""One-rank RCCL sanity on the 1-GPU box: the backend loads, and the collectives dist.py uses
(all_to_all_single with split sizes, all_reduce, broadcast, all_gather_into_tensor) run on device tensors."""
import torch
import torch.distributed as dist

dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29411", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.arange(8, dtype=torch.float64, device="cuda")
out = torch.empty_like(t)
w = dist.all_to_all_single(out, t, output_split_sizes=[8], input_split_sizes=[8], async_op=True)
w.wait()
dist.all_reduce(t)
dist.broadcast(t, 0)
g = torch.empty(8, dtype=torch.float64, device="cuda")
dist.all_gather_into_tensor(g, t)
torch.cuda.synchronize()
assert torch.equal(out, torch.arange(8, dtype=torch.float64, device="cuda")) and torch.equal(g, t)
print("rccl ok:", torch.cuda.get_device_name(0), "nccl", torch.cuda.nccl.version())
dist.destroy_process_group()
