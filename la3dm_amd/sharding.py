"""Block sharding of one packed scan across GPUs (BASELINE config 5).

Test blocks are independent given the per-scan training set, so a scan shards with no
data-path collective: rank r takes every `world`-th test block of the (heaviest-first) packed
order — near-perfect balance, and each rank's list stays heaviest-first.  The training CSR is
replicated.  After the kernel, ONE all-gather of the per-rank leaf arrays (alpha | beta | state,
padded to the largest shard) gives every rank the whole updated grid; `reassemble` scatters it
back into the packed order, after which BGKOctoMap.commit() writes the nodes and prunes.

Pure index bookkeeping (numpy); the collective itself is torch.distributed (RCCL on GPUs,
gloo in the CPU tests).
"""
import os

import numpy as np


class Shard:
    """The slice of a packed scan owned by one rank (numpy arrays, C-contiguous)."""

    def __init__(self, pk, rank, world):
        self.rank, self.world = rank, world
        nt = pk.n_test_blk
        self.blocks = np.arange(rank, nt, world, dtype=np.int64)      # indices into the packed order
        lo = pk.leaf_off.astype(np.int64)
        cnt = lo[self.blocks + 1] - lo[self.blocks]
        self.leaf_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
        # global leaf index of every local leaf
        self.leaf_index = (np.repeat(lo[self.blocks] - self.leaf_off[:-1].astype(np.int64), cnt)
                           + np.arange(int(cnt.sum()), dtype=np.int64))
        self.n_test_blk = int(self.blocks.size)
        self.n_leaf = int(cnt.sum())
        self.nbr = np.ascontiguousarray(pk.nbr[self.blocks])
        self.blk_center = np.ascontiguousarray(pk.blk_center[self.blocks])
        self.leaf_key = np.ascontiguousarray(pk.leaf_key[self.leaf_index])
        self.alpha = np.ascontiguousarray(pk.alpha[self.leaf_index])
        self.beta = np.ascontiguousarray(pk.beta[self.leaf_index])
        self.state = np.zeros(self.n_leaf, np.uint8)
        # replicated
        self.train_xyzy, self.train_off = pk.train_xyzy, pk.train_off
        self.n_train_pts, self.n_train_blk = pk.n_train_pts, pk.n_train_blk
        self.flags = pk.flags


def shard_leaf_counts(pk, world):
    lo = pk.leaf_off.astype(np.int64)
    cnt = np.diff(lo)
    return [int(cnt[r::world].sum()) for r in range(world)]


def pack_payload(alpha, beta, state, cap):
    """alpha | beta | state of one shard as a byte vector padded to `cap` leaves (9 bytes/leaf)."""
    out = np.zeros(9 * cap, np.uint8)
    n = alpha.size
    out[0:4 * n] = alpha.view(np.uint8)
    out[4 * cap:4 * cap + 4 * n] = beta.view(np.uint8)
    out[8 * cap:8 * cap + n] = state
    return out


def reassemble(pk, gathered, world):
    """Scatter the all-gathered payloads ([world, 9*cap] bytes) into pk.alpha/beta/state."""
    gathered = np.asarray(gathered, np.uint8).reshape(world, -1)
    cap = gathered.shape[1] // 9
    lo = pk.leaf_off.astype(np.int64)
    for r in range(world):
        blocks = np.arange(r, pk.n_test_blk, world, dtype=np.int64)
        cnt = lo[blocks + 1] - lo[blocks]
        n = int(cnt.sum())
        start = np.concatenate([[0], np.cumsum(cnt)])[:-1]
        idx = np.repeat(lo[blocks] - start, cnt) + np.arange(n, dtype=np.int64)
        row = gathered[r]
        pk.alpha[idx] = row[0:4 * n].view(np.float32)
        pk.beta[idx] = row[4 * cap:4 * cap + 4 * n].view(np.float32)
        pk.state[idx] = row[8 * cap:8 * cap + n]


# ---- device-resident sharded insert (la3dm_devmap_set_shard): the all-gather callback on torch.distributed ----
class _DeviceBytes:
    """a raw device pointer as a __cuda_array_interface__ object, so torch can view it without a copy"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3}


def exchange_v(dist, segments, r, world, group=None):
    """ONE grouped exchange for a whole all-gather-v (VERDICT r04 #4c): `segments` = [(buf, offsets, nbytes), ...] — byte
    tensors (host or device) of which rank q owns [offsets[q], offsets[q] + nbytes[q]).  Every rank sends each of its ranges
    to every other rank and receives theirs, all of it as one torch.distributed.batch_isend_irecv: on the "nccl" backend
    (RCCL) that is a single ncclGroupStart / ncclGroupEnd — one fused launch per exchange whatever the number of arrays and
    ranks (before: `world` broadcasts per array, 24 collectives per insert at 8 ranks), and on xGMI's point-to-point links
    every pair of GPUs talks over its own link, once.  Sends and receives between a pair are listed in the same (segment)
    order on both sides, which is what NCCL / gloo match them by.  Blocks the host until the local requests complete (for
    NCCL: until they are queued on the current stream)."""
    if os.environ.get("LA3DM_SHARD_EXCHANGE", "p2p") == "broadcast":
        # diagnosis switch: the round-3 / round-4 form — one broadcast per rank and array, issued one by one
        works = []
        for buf, offsets, nbytes in segments:
            for q in range(world):
                if nbytes[q]:
                    src = dist.get_global_rank(group, q) if group is not None else q
                    works.append(dist.broadcast(buf[offsets[q]:offsets[q] + nbytes[q]], src=src, group=group, async_op=True))
        for w in works:
            if w is not None:
                w.wait()
        return
    ops = []
    for buf, offsets, nbytes in segments:
        mine = buf[offsets[r]:offsets[r] + nbytes[r]]
        for q in range(world):
            if q == r:
                continue
            peer = dist.get_global_rank(group, q) if group is not None else q
            if nbytes[r]:
                ops.append(dist.P2POp(dist.isend, mine, peer, group))
            if nbytes[q]:
                ops.append(dist.P2POp(dist.irecv, buf[offsets[q]:offsets[q] + nbytes[q]], peer, group))
    if not ops:
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def gather_v(dist, buf, offsets, nbytes, r, world, group=None):
    """in-place all-gather-v of ONE byte buffer (kept for callers with a single array): exchange_v on one segment.
    Blocking like exchange_v (round 5 dropped the `async_op` form: there is no list of work handles any more)."""
    exchange_v(dist, [(buf, offsets, nbytes)], r, world, group=group)


def leaf_segments(leaf_bounds, keys=True):
    """host mirror of what la3dm_devmap_insert_* hands the callback: (offsets, nbytes) per rank for the alpha, beta (4 B per
    leaf), state (1 B per leaf) and — single-pass scans since round 5: a rank lists only its own range's leaves, the others'
    arrive with their keys — leaf-key (4 B per leaf) arrays, from the leaf index bounds [world + 1] of the ranks' ranges:
    13 B per leaf of the scan"""
    lb = [int(x) for x in leaf_bounds]
    n = [lb[q + 1] - lb[q] for q in range(len(lb) - 1)]
    four = ([4 * x for x in lb[:-1]], [4 * x for x in n])
    return [four, four, (lb[:-1], n)] + ([four] if keys else [])


def torch_allgather(dist, rank, device, stage_through_host=False, group=None):
    """-> allgatherv(segments, world, rank, stream) for BGKOctoMap.set_shard: the in-place all-gather-v of the scan's leaf
    arrays on torch.distributed (RCCL over xGMI when the process group's backend is "nccl"), queued on the map's own HIP
    stream (wrapped as a torch ExternalStream: the collective waits for what the library queued before it, what the
    library queues after it waits for the collective; no host synchronisation).  Rank q owns a different number of leaves
    in general (an all-gather-v): all arrays of an exchange go out as ONE grouped batch of point-to-point sends / receives
    (exchange_v: a single ncclGroupStart / End on RCCL).
    stage_through_host: the gloo self-test on a single GPU (all ranks on cuda:0) — the payload goes through pinned host
    mirrors of the segments, exchanged by the same exchange_v (one grouped batch of sends / receives) as on RCCL, with the
    synchronisations that needs; never a measurement.
    group: the process group the shard ranks 0 .. world - 1 are the members of (default: the default group); the shard
    rank handed to set_shard must be the rank INSIDE that group."""
    import torch

    def allgatherv(segments, world, r, stream):
        ext = torch.cuda.ExternalStream(stream, device=device) if stream else torch.cuda.current_stream(device)
        with torch.cuda.stream(ext):
            segs = []
            for base, offsets, nbytes in segments:
                end = max(o + n for o, n in zip(offsets, nbytes))
                if end == 0:
                    continue
                segs.append((torch.as_tensor(_DeviceBytes(base, end), device=device), offsets, nbytes))
            if not segs:
                return
            if not stage_through_host:
                exchange_v(dist, segs, r, world, group=group)   # (NCCL: queued on `ext`; the host is not blocked by the transfer)
                return
            # single-GPU self-test (gloo cannot move device memory): the SAME grouped exchange on pinned host mirrors of the
            # segments — own range down, exchange_v, the others' ranges up — with the synchronisations that needs
            ext.synchronize()
            host = []
            for buf, offsets, nbytes in segs:
                h = torch.empty(buf.numel(), dtype=torch.uint8).pin_memory()
                if nbytes[r]:
                    h[offsets[r]:offsets[r] + nbytes[r]].copy_(buf[offsets[r]:offsets[r] + nbytes[r]])
                host.append((h, offsets, nbytes))
            ext.synchronize()
            exchange_v(dist, host, r, world, group=group)
            for (buf, offsets, nbytes), (h, _, _) in zip(segs, host):
                for q in range(world):
                    if q != r and nbytes[q]:
                        buf[offsets[q]:offsets[q] + nbytes[q]].copy_(h[offsets[q]:offsets[q] + nbytes[q]])
            ext.synchronize()

    return allgatherv


def layer_cuts(hist, world):
    """host mirror of shard_layer_cuts (la3dm_amd/csrc/devmap.hip): the sharded sample filter cuts the z-layers of the filter
    grid into `world` contiguous ranges — layer j opens rank q's range when the running sample count has reached q / world
    of the total.  -> cut [world + 1] (layer indices)"""
    hist = [int(x) for x in hist]
    total, n = sum(hist), len(hist)
    cut = [0] + [n] * world
    run, q = 0, 1
    for j in range(n):
        if q >= world:
            break
        while q < world and run * world >= total * q:
            cut[q] = j
            q += 1
        run += hist[j]
    return np.array(cut, np.int64)


def balanced_ranges(weights, world):
    """host mirror of dm_shard_weight / dm_shard_bounds (devmap_kernels.h): cut the test-block list (candidate order)
    into `world` contiguous ranges where the running weight (neighbourhood size + 16 per block) crosses q / world of the
    total.  -> bounds [world + 1]"""
    w = np.asarray(weights, np.uint64) + 16
    if len(w):
        w = np.minimum(w, np.uint64(max(32, (1 << 31) // len(w))))
    cum = np.concatenate([[0], np.cumsum(w)[:-1]]).astype(np.uint64)          # exclusive
    total = int(cum[-1] + w[-1]) if len(w) else 0   # (the device caps a block's weight at 2^31 / n so that its 32-bit sum cannot wrap)
    b = [int(np.searchsorted(cum, np.uint64(total * q // world), side="left")) for q in range(world)]
    return np.array(b + [len(w)], np.int64)
