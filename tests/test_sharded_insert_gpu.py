"""Block-sharded, device-resident insert_pointcloud: `world` replicas of the map,
every rank predicts + fuses a contiguous range of the test blocks (src/bgkoctomap/bgkoctomap.cpp:293-336 is the loop
that is cut), one in-place all-gather-v of the leaves, commit + prune everywhere; the samples' voxel filter of the front end is
divided over the ranks by z-layer of its grid and its output all-gathered.  Bar: every replica ends up BIT-IDENTICAL to
the map a single process builds from the same clouds — two fused scans, so the second one runs on a pruned pool.
The ranks share the one GPU of the test box and exchange over gloo — through the production sharding.exchange_v (one grouped
batch of sends / receives per exchange) on pinned host mirrors of the segments; with RCCL only the transport changes."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single(variant, rays):
    import la3dm_amd
    if variant == "gp":
        m, fr = la3dm_amd.GPOctoMap(**la3dm_amd.GP_YAML, device=0), 0.1
    else:
        m, fr = la3dm_amd.BGKOctoMap(**dict(la3dm_amd.BGK_YAML, block_depth=int(variant[1:])), device=0), 0.5
    for pose in (None, (1.5, 0.5, 1.0)):
        xyz, origin = la3dm_amd.synthetic_scan(rays, origin=pose)
        m.insert_pointcloud(xyz, origin, 0.1, fr, -1.0)
    return m.leaves(), int(m.stats()["voxel_updates"]), m.training_data()


@pytest.mark.parametrize("variant,rays,world,sum_mode,slab", [("d3", 30000, 2, "0", "1"), ("d4", 20000, 3, "0", "1"), ("gp", 8000, 2, "0", "1"),
                                                               ("d3", 30000, 2, "1", "1"), ("d4", 20000, 3, "1", "1"), ("d3", 30000, 3, "0", "0"),
                                                               ("gp", 8000, 3, "0", "1")])
def test_sharded_replicas_equal_the_single_process_map(built, tmp_path, variant, rays, world, sum_mode, slab, monkeypatch):
    """sum_mode "1" is the library's default accumulate mode (double sums; table kernel on the un-pruned blocks, general
    kernel on the pruned ones of the second scan): a leaf's sums are formed on exactly one rank by the same kernel as in
    the single process, so the replicas are bit-identical there too.
    slab "1" (the default, round 6): the x-slab partition — every rank forms the per-block point counts of the whole scan but sorts
    membership pairs and builds the CSR, the training rows and the neighbour tables for the blocks of its own range of the test list
    (+ halo) only (devmap_kernels.h "x-slab partition"; src/bgkoctomap/bgkoctomap.cpp:234-284 is the loop that is divided);
    "0" (LA3DM_SHARD_SLAB=0): the CSR of all training blocks on every rank, as before."""
    monkeypatch.setenv("LA3DM_BGK_SUM", sum_mode)          # the single-process reference below; the workers inherit env
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", LA3DM_DEBUG_SHARD="1", LA3DM_BGK_SUM=sum_mode,
               LA3DM_SHARD_SLAB=slab)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + (os.getpid() % 500)), os.path.join(ROOT, "tests", "helpers", "shard_worker.py"),
           str(tmp_path), variant, str(rays)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for rank in range(world):        # the sharded form of the samples' filter really ran, on every rank
        assert f"sharded sample filter: rank {rank} of {world}" in r.stderr, r.stderr[-2000:]
        assert (f"x-slab partition: rank {rank} of {world}" in r.stderr) == (slab == "1"), r.stderr[-2000:]
    if slab == "1":                  # ... and a rank's CSR holds a fraction of the scan's membership pairs, not all of them
        import re
        pairs = {}
        for mm in re.finditer(r"x-slab partition: rank (\d+) of \d+ builds the CSR of (\d+) \(block, point\) pairs", r.stderr):
            pairs.setdefault(int(mm.group(1)), []).append(int(mm.group(2)))
        last = [pairs[q][-1] for q in range(world)]
        assert max(last) < 0.9 * sum(last), last
    ref, U, train = _single(variant, rays)
    for rank in range(world):
        got = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        assert int(got["voxel_updates"]) == U
        # the samples' voxel filter is divided over the ranks by z-layer and all-gathered (devmap_kernels.h "sharded sample
        # filter"): every rank must hold the single-process training set of the last scan, point for point, in its order
        assert got["training"].shape == train.shape, (rank, got["training"].shape, train.shape)
        assert (got["training"].view(np.uint32) == train.view(np.uint32)).all(), rank
        assert got["block_key"].size == ref["block_key"].size, rank
        for k in ("block_key", "node_key", "state", "classified"):
            assert (got[k] == ref[k]).all(), (rank, k)
        for k in ("A", "B"):
            assert (got[k].view(np.uint32) == ref[k].view(np.uint32)).all(), (rank, k)


@pytest.mark.parametrize("world,failing", [(2, 1), (3, 0)])
def test_a_rank_local_failure_ends_the_insert_on_every_rank(built, tmp_path, world, failing):
    """a rank that fails on its own ahead of the front end's collectives (out of memory, a tripped
    scan, ...) must not leave its peers waiting in them.  LA3DM_INJECT_FRONT_END_FAILURE makes rank `failing` fail before
    it has done anything; the sharded front end's status exchange (entered by every rank whatever happened to it) makes
    all ranks give the insert up: the failed rank reports its own error, every other rank LA3DM_ERR_PEER naming it —
    and the job ends (no hang: the subprocess timeout would catch it)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", LA3DM_INJECT_FRONT_END_FAILURE=str(failing))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + (os.getpid() % 500)), os.path.join(ROOT, "tests", "helpers", "shard_worker.py"),
           str(tmp_path), "d3", "8000"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for rank in range(world):
        msg = open(os.path.join(tmp_path, f"rank{rank}.err")).read()
        assert msg != "NO ERROR", rank
        if rank == failing:
            assert "injected rank-local front-end failure" in msg, (rank, msg)
        else:
            assert f"rank {failing} failed in its front end" in msg, (rank, msg)


@pytest.mark.parametrize("world,failing", [(2, 0), (3, 2)])
def test_a_failure_after_the_cut_does_not_hang_the_peers(built, tmp_path, world, failing):
    """A rank that fails AFTER the range cut (round 6: in the x-slab partition — out of memory for its membership pairs, say;
    LA3DM_INJECT_SLAB_FAILURE) has peers that are about to wait in the leaf exchange: it must still enter it.  It skips its kernel,
    hands its range over as "update() ran on no leaf" (the write-back skips such leaves on every replica) and reports its own error;
    the other ranks finish the insert — and the job ends (no hang: the subprocess timeout would catch it)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", LA3DM_INJECT_SLAB_FAILURE=str(failing))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + (os.getpid() % 500)), os.path.join(ROOT, "tests", "helpers", "shard_worker.py"),
           str(tmp_path), "d3", "8000"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for rank in range(world):
        msg = open(os.path.join(tmp_path, f"rank{rank}.err")).read()
        if rank == failing:
            assert "injected rank-local failure in the x-slab partition" in msg, (rank, msg)
        else:
            assert msg == "NO ERROR", (rank, msg)


def test_allgather_callback_on_rccl_single_rank(built):
    """The transport the driver's multi-GPU runs use is torch.distributed's nccl backend (RCCL); the box these tests run on
    has one GPU, so the sharded tests above exchange over gloo (the same sharding.exchange_v, on pinned host mirrors).  This
    runs what one GPU can of the production transport (tests/helpers/rccl_single_rank.py): the production callback on a
    ONE-rank nccl group (no peers: plumbing only, it issues no operation) and then REAL RCCL operations — all-gather,
    broadcast, all-reduce and, where accepted, a grouped send / receive to self — on the map-stream-as-ExternalStream with
    work queued before and after them."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_single_rank.py")],
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rccl single-rank ok: nccl" in r.stdout
    assert "all_gather_into_tensor, broadcast, all_reduce" in r.stdout, r.stdout[-500:]
