"""worker of tests/test_sharded_insert_gpu.py::test_allgather_callback_on_rccl_single_rank: what a ONE-GPU box can run of the
production transport — torch.distributed's "nccl" backend (= RCCL on ROCm) with one rank.

1. the production (non-staged) callback la3dm_amd.sharding.torch_allgather on that group: process-group setup, the raw device
   pointer wrapped through __cuda_array_interface__, the map's stream as a torch ExternalStream.  With one rank its exchange
   has no peers, so sharding.exchange_v issues NO operation here (its send / receive list is empty) — this part checks the
   plumbing and that the buffer is left intact, nothing more.
2. REAL RCCL operations queued on that same ExternalStream, ordered against work queued on it before and after, on uint8
   views of raw device pointers like the callback's: all_gather_into_tensor (world = 1: a device copy done by RCCL's
   kernel), broadcast from rank 0, all_reduce(SUM), and — where this torch / RCCL accepts a rank sending to itself — one
   grouped batch_isend_irecv (ncclGroupStart / ncclSend + ncclRecv / ncclGroupEnd), the call exchange_v makes with peers.
   The line it prints says which of them ran."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import torch
    import torch.distributed as dist
    from la3dm_amd import sharding
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29650 + os.getpid() % 200))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    fn = sharding.torch_allgather(dist, 0, dev)
    side = torch.cuda.Stream(device=dev)                      # stands for the map's own HIP stream
    with torch.cuda.stream(side):
        buf = torch.arange(4096, dtype=torch.int32, device=dev).view(torch.uint8)
        ref = buf.clone()
    side.synchronize()
    # 1. the callback: one rank owns everything, nothing to exchange
    fn([(buf.data_ptr(), [0], [buf.numel()])], 1, 0, side.cuda_stream)
    fn([(buf.data_ptr(), [12], [1200]), (buf.data_ptr() + 8192, [0], [4])], 1, 0, side.cuda_stream)
    side.synchronize()
    assert bool((buf == ref).all())
    # 2. real RCCL operations on the ExternalStream
    ran = []
    ext = torch.cuda.ExternalStream(side.cuda_stream, device=dev)
    with torch.cuda.stream(ext):
        src = torch.as_tensor(sharding._DeviceBytes(buf.data_ptr(), buf.numel()), device=dev)
        big = torch.empty(1 << 24, dtype=torch.uint8, device=dev)
        big.copy_(src.repeat((1 << 24) // src.numel()))       # queued on the stream BEFORE the collective: it must wait for it
        out = torch.zeros_like(big)
        dist.all_gather_into_tensor(out, big)
        ran.append("all_gather_into_tensor")
        check = (out == big).all()                            # queued AFTER: it must see the gathered bytes
        b2 = src.clone()
        dist.broadcast(b2, src=0)
        ran.append("broadcast")
        cnt = torch.full((8,), 3, dtype=torch.int32, device=dev)
        dist.all_reduce(cnt)
        ran.append("all_reduce")
        try:
            if os.environ.get("LA3DM_TEST_RCCL_SELF_P2P", "1") != "1":
                raise RuntimeError("switched off (LA3DM_TEST_RCCL_SELF_P2P=0)")
            rx = torch.zeros(4096, dtype=torch.uint8, device=dev)
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, src[:4096], 0), dist.P2POp(dist.irecv, rx, 0)]):
                req.wait()
            ext.synchronize()
            assert bool((rx == src[:4096]).all())
            ran.append("batch_isend_irecv(self)")
        except Exception as e:                                # noqa: BLE001  (send-to-self is refused by some torch versions)
            print("batch_isend_irecv to self not available here:", type(e).__name__, str(e)[:120])
    ext.synchronize()
    assert bool(check) and bool((b2 == src).all()) and bool((cnt == 3).all())
    print("rccl single-rank ok:", dist.get_backend(), torch.cuda.get_device_name(0), "| ran on RCCL:", ", ".join(ran))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
