"""worker of tests/test_sharded_insert_gpu.py: one rank of a block-sharded, device-resident insert_pointcloud.
All ranks share cuda:0 (single-GPU box) and exchange the leaf payload over gloo, staged through host memory — the
protocol, the range cut, pack / all-gather / unpack and commit are the production code; only the transport differs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    out_dir, variant, rays = sys.argv[1], sys.argv[2], int(sys.argv[3])
    import torch
    import torch.distributed as dist
    import la3dm_amd
    from la3dm_amd import sharding
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if variant == "gp":
        m = la3dm_amd.GPOctoMap(**la3dm_amd.GP_YAML, device=0)
        fr = 0.1
    else:
        m = la3dm_amd.BGKOctoMap(**dict(la3dm_amd.BGK_YAML, block_depth=int(variant[1:])), device=0)
        fr = 0.5
    m.set_shard(rank, world, sharding.torch_allgather(dist, rank, dev, stage_through_host=True))
    if os.environ.get("LA3DM_INJECT_FRONT_END_FAILURE") is not None or os.environ.get("LA3DM_INJECT_SLAB_FAILURE") is not None:
        # one rank fails on its own in the front end: EVERY rank's insert must come back with an error (none may wait in a
        # collective the failed rank never enters); the message goes to the test
        xyz, origin = la3dm_amd.synthetic_scan(rays)
        try:
            m.insert_pointcloud(xyz, origin, 0.1, fr, -1.0)
            msg = "NO ERROR"
        except Exception as e:                                   # noqa: BLE001
            msg = f"{type(e).__name__}: {e}"
        with open(os.path.join(out_dir, f"rank{rank}.err"), "w") as f:
            f.write(msg)
        dist.barrier()
        dist.destroy_process_group()
        return
    for pose in (None, (1.5, 0.5, 1.0)):
        xyz, origin = la3dm_amd.synthetic_scan(rays, origin=pose)
        m.insert_pointcloud(xyz, origin, 0.1, fr, -1.0)
    lv = m.leaves()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{k: lv[k] for k in ("block_key", "node_key", "A", "B", "state", "classified")},
             voxel_updates=np.int64(m.stats()["voxel_updates"]), training=m.training_data())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
