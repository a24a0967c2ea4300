"""CPU coverage of the device-resident sharded insert's host-visible logic (la3dm_amd/sharding.py): the range cut that
mirrors dm_shard_bounds, and the payload exchange protocol (pack -> ONE in-place all-gather -> unpack) driven over a
2-rank gloo group with the device step emulated in numpy."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_balanced_ranges_cover_the_list_once_and_balance_the_weight():
    from la3dm_amd import sharding
    rng = np.random.default_rng(3)
    for n, world in ((1, 2), (5, 8), (1000, 2), (41694, 8), (261000, 8)):
        w = rng.integers(0, 400, n)
        b = sharding.balanced_ranges(w, world)
        assert b[0] == 0 and b[-1] == n and (np.diff(b) >= 0).all()
        if n >= 100 * world:
            per = np.array([(w[b[q]:b[q + 1]] + 16).sum() for q in range(world)], np.float64)
            assert per.max() / per.mean() < 1.05
    assert (sharding.balanced_ranges(np.zeros(0), 4) == 0).all()


def _rank(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from la3dm_amd import sharding
    rng = np.random.default_rng(11)                      # same "scan" on every rank
    n_test = 1234
    nleaf = rng.integers(1, 65, n_test)
    leaf_off = np.concatenate([[0], np.cumsum(nleaf)])
    bounds = sharding.balanced_ranges(rng.integers(0, 300, n_test), world)
    lb = leaf_off[bounds]
    chunk = int(((np.diff(lb).max() + 63) // 64) * 64)
    truth_a = rng.random(leaf_off[-1]).astype(np.float32)
    truth_b = rng.random(leaf_off[-1]).astype(np.float32)
    truth_s = rng.integers(0, 256, leaf_off[-1]).astype(np.uint8)
    a, b, s = np.zeros_like(truth_a), np.zeros_like(truth_b), np.zeros_like(truth_s)
    lo, hi = lb[rank], lb[rank + 1]                      # "predict + fuse" of this rank's range only
    a[lo:hi], b[lo:hi], s[lo:hi] = truth_a[lo:hi], truth_b[lo:hi], truth_s[lo:hi]
    payload = np.zeros(9 * chunk * world, np.uint8)      # dm_shard_pack
    sl = payload[9 * chunk * rank:9 * chunk * (rank + 1)]
    sl[:4 * (hi - lo)] = a[lo:hi].view(np.uint8)
    sl[4 * chunk:4 * chunk + 4 * (hi - lo)] = b[lo:hi].view(np.uint8)
    sl[8 * chunk:8 * chunk + (hi - lo)] = s[lo:hi]
    t = torch.from_numpy(payload)
    dist.all_gather_into_tensor(t, t[9 * chunk * rank:9 * chunk * (rank + 1)].clone())   # the ONE collective
    for q in range(world):                               # dm_shard_unpack
        if q == rank:
            continue
        n = lb[q + 1] - lb[q]
        sq = payload[9 * chunk * q:9 * chunk * (q + 1)]
        a[lb[q]:lb[q + 1]] = sq[:4 * n].view(np.float32)
        b[lb[q]:lb[q + 1]] = sq[4 * chunk:4 * chunk + 4 * n].view(np.float32)
        s[lb[q]:lb[q + 1]] = sq[8 * chunk:8 * chunk + n]
    ret[rank] = bool((a == truth_a).all() and (b == truth_b).all() and (s == truth_s).all())
    dist.destroy_process_group()


def test_payload_protocol_two_ranks_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29800 + os.getpid() % 100
    procs = [mp.get_context("spawn").Process(target=_rank, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(world))
