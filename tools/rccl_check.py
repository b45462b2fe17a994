import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
part = torch.arange(3 * 37, dtype=torch.float64, device=dev).view(3, 37)
parts = torch.empty((1, 3, 37), dtype=torch.float64, device=dev)
dist.all_gather_into_tensor(parts, part.contiguous())
assert torch.equal(parts[0], part)
flat = torch.ones(1000, device=dev); dist.all_reduce(flat); assert float(flat.sum()) == 1000
dist.barrier(); dist.destroy_process_group(); print("rccl one-rank collectives ok")
