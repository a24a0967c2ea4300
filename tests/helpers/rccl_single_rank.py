"""worker of tests/test_sharded_insert_gpu.py::test_allgather_callback_on_rccl_single_rank: the production (non-staged)
form of la3dm_amd.sharding.torch_allgather on the nccl (= RCCL) backend with ONE rank — the most the single-GPU test box
allows: process-group setup, the raw device pointer wrapped through __cuda_array_interface__, the map's stream as a
torch ExternalStream, all_gather_into_tensor (even ranges) and broadcast (uneven ranges) on uint8 views, stream-ordered
completion."""
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
    # even ranges -> all_gather_into_tensor; a range that does not start at 0 -> the broadcast form
    fn([(buf.data_ptr(), [0], [buf.numel()])], 1, 0, side.cuda_stream)
    fn([(buf.data_ptr(), [12], [1200]), (buf.data_ptr() + 8192, [0], [4])], 1, 0, side.cuda_stream)
    side.synchronize()
    assert bool((buf == ref).all())
    print("rccl single-rank ok:", dist.get_backend(), torch.cuda.get_device_name(0))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
