"""CPU coverage of the device-resident sharded insert's host-visible logic (la3dm_amd/sharding.py): the range cut that
mirrors dm_shard_bounds, and the exchange protocol (ONE in-place all-gather-v over the alpha / beta / state arrays: a rank's
leaves are a contiguous range of each) driven over 2- and 3-rank gloo groups with the device step emulated in numpy."""
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


def test_layer_cuts_of_the_sharded_sample_filter():
    """the z-layer cut of the samples' voxel filter (mirror of shard_layer_cuts): contiguous, monotone, covers every layer once,
    balanced to within the heaviest layer"""
    from la3dm_amd import sharding
    rng = np.random.default_rng(5)
    for n, world in ((1, 2), (3, 8), (80, 2), (80, 8), (700, 8), (64, 3)):
        hist = rng.integers(0, 5000, n)
        hist[n // 3] += 200000                     # the sensor's layer holds a sample of every beam
        cut = sharding.layer_cuts(hist, world)
        assert cut[0] == 0 and cut[-1] == n and (np.diff(cut) >= 0).all()
        per = np.array([hist[cut[q]:cut[q + 1]].sum() for q in range(world)], np.float64)
        assert per.sum() == hist.sum()
        if n >= 8 * world:
            assert per.max() <= hist.sum() / world + hist.max()
    assert (sharding.layer_cuts(np.zeros(10, np.int64), 4)[1:-1] == 0).all()


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
    truth_a = rng.random(leaf_off[-1]).astype(np.float32)
    truth_b = rng.random(leaf_off[-1]).astype(np.float32)
    truth_s = rng.integers(0, 256, leaf_off[-1]).astype(np.uint8)
    truth_k = rng.integers(0, 1 << 18, leaf_off[-1]).astype(np.uint32)   # leaf keys: only the owner lists its range's leaves
    a, b, s, k = np.zeros_like(truth_a), np.zeros_like(truth_b), np.zeros_like(truth_s), np.zeros_like(truth_k)
    lo, hi = lb[rank], lb[rank + 1]                      # "predict + fuse" of this rank's range only
    a[lo:hi], b[lo:hi], s[lo:hi], k[lo:hi] = truth_a[lo:hi], truth_b[lo:hi], truth_s[lo:hi], truth_k[lo:hi]
    # the exchange: ONE in-place all-gather-v over the four leaf arrays (no pack / unpack, no padding), exactly the
    # segments la3dm_devmap_insert_* hands the callback
    # all four arrays in ONE grouped batch of sends / receives (sharding.exchange_v: what torch_allgather issues per exchange)
    segs = [(torch.from_numpy(arr.view(np.uint8)), offsets, nbytes) for arr, (offsets, nbytes) in zip((a, b, s, k), sharding.leaf_segments(lb))]
    sharding.exchange_v(dist, segs, rank, world)
    ret[rank] = bool((a == truth_a).all() and (b == truth_b).all() and (s == truth_s).all() and (k == truth_k).all())
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 3])
def test_payload_protocol_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29800 + (os.getpid() + 7 * world) % 100
    procs = [mp.get_context("spawn").Process(target=_rank, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(world))


def _rank_subgroup(rank, port, ret):
    """world of 3, the shard group is global ranks {1, 2}: shard rank q is NOT global rank q (broadcast's src
    is a global rank)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=3)
    sys.path.insert(0, ROOT)
    from la3dm_amd import sharding
    grp = dist.new_group([1, 2])                          # (every rank of the default group makes the call)
    ok = True
    if rank in (1, 2):
        r, world = rank - 1, 2
        truth = np.arange(1000, dtype=np.uint8)
        buf = np.zeros_like(truth)
        offsets, nbytes = [0, 300], [300, 700]            # uneven ranges
        buf[offsets[r]:offsets[r] + nbytes[r]] = truth[offsets[r]:offsets[r] + nbytes[r]]
        sharding.gather_v(dist, torch.from_numpy(buf), offsets, nbytes, r, world, group=grp)
        ok = bool((buf == truth).all())
        buf2 = np.zeros(1000, np.uint8)                   # even ranges, and a rank that owns nothing in a second array
        buf2[500 * r:500 * (r + 1)] = truth[500 * r:500 * (r + 1)]
        buf3 = np.zeros(64, np.uint8)
        if r == 1:
            buf3[:] = 7
        sharding.exchange_v(dist, [(torch.from_numpy(buf2), [0, 500], [500, 500]), (torch.from_numpy(buf3), [0, 0], [0, 64])], r, world, group=grp)
        ok = ok and bool((buf2 == truth).all()) and bool((buf3 == 7).all())
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_payload_protocol_on_a_subgroup_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29900 + os.getpid() % 90
    procs = [mp.get_context("spawn").Process(target=_rank_subgroup, args=(r, port, ret)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(3))
