import os, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
x = torch.full((4,), float(rank + 1), device="cuda", dtype=torch.float64)
out = [torch.zeros(4, device="cuda", dtype=torch.float64) for _ in range(world)]
dist.all_gather(out, x)
torch.cuda.synchronize()
print(rank, [o[0].item() for o in out], flush=True)
dist.destroy_process_group()
