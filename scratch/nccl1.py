import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
a = [torch.full((1 << 20,), i, dtype=torch.uint8, device=dev) for i in (1, 2)]
o = [torch.zeros(1 << 20, dtype=torch.uint8, device=dev) for _ in range(2)]
pend = [None, None]
for k in range(6):
    b = k & 1
    if pend[b] is not None: pend[b].wait()
    a[b].add_(1)
    pend[b] = dist.all_gather_into_tensor(o[b], a[b], async_op=True)
for w in pend: w.wait()
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
print("ok", int(o[0][0]), int(o[1][0]))
dist.destroy_process_group()
