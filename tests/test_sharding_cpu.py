"""Multi-process path on CPU (gloo, world_size 2): block sharding of one scan, the all-gather
of the per-rank leaf arrays and the reassembly/commit give exactly the single-process map.
The device step is emulated per shard with the oracle's predict (the oracle is the checker)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, pcd_path

YAML = dict(resolution=0.1, block_depth=3, sf2=1.0, ell=0.2, free_thresh=0.3, occupied_thresh=0.7, var_thresh=100.0,
            prior_A=0.001, prior_B=0.001)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import la3dm_amd
    from la3dm_amd import sharding
    from test_host_logic import _emulate_device
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = la3dm_amd.BGKOctoMap(**YAML, device=-1)
    for i in (1, 2):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        assert m.prepare(xyz, origin, 0.1, 0.5, 8.0)
        pk = m.packed()
        sh = sharding.Shard(pk, rank, world)
        _emulate_device(sh, YAML)                       # this rank's blocks only
        cap = max(sharding.shard_leaf_counts(pk, world))
        mine = torch.from_numpy(sharding.pack_payload(sh.alpha, sh.beta, sh.state, cap))
        out = torch.zeros(world * 9 * cap, dtype=torch.uint8)
        dist.all_gather_into_tensor(out, mine)
        sharding.reassemble(pk, out.numpy(), world)
        m.commit()
    lv = m.leaves()
    q.put((rank, {k: v.copy() for k, v in lv.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_scan_equals_single_process(built):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    import la3dm_amd
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    o = O.OracleMap(**YAML)
    for i in (1, 2):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    ref = o.leaves()
    for r in (0, 1):
        for k in ("block_key", "node_key", "A", "B", "state", "classified"):
            assert (res[r][k] == ref[k]).all(), (r, k)


def test_shard_bookkeeping(built):
    import la3dm_amd
    from la3dm_amd import sharding
    m = la3dm_amd.BGKOctoMap(**YAML, device=-1)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 3))
    m.prepare(xyz, origin, 0.1, 0.5, 8.0)
    pk = m.packed()
    for world in (1, 2, 3, 8):
        shards = [sharding.Shard(pk, r, world) for r in range(world)]
        allidx = np.concatenate([s.leaf_index for s in shards])
        assert np.sort(allidx).tolist() == list(range(pk.n_leaf))          # a partition of the leaves
        assert sum(s.n_test_blk for s in shards) == pk.n_test_blk
        w = [sum(int(pk.train_off[n + 1] - pk.train_off[n]) for n in s.nbr.ravel() if n >= 0) for s in shards]
        assert max(w) - min(w) <= 0.05 * max(w) + 400                       # balanced by training points
        assert sharding.shard_leaf_counts(pk, world) == [s.n_leaf for s in shards]
