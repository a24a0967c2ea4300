"""The device-resident front end's own primitives against numpy: the single-launch exclusive scan (plain and "head flags
of a sorted key array" forms, la3dm_amd/csrc/devmap_scan.h) and the one-launch-per-pass stable radix sort
(devmap_sort.h), through their C-ABI test hooks.  Sizes straddle the tile sizes (4096), the sort's tile groups and shapes, the resident / ticket
switch (more than 1024 / 512 tiles) and run back to back on one map, so that every launch starts on the state the previous one
left behind (the status arrays and histograms clean themselves)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["occupancy", "4 workgroups"])
def devmap(built, request):
    """second run: LA3DM_SCAN_RESIDENT / LA3DM_RADIX_RESIDENT = 4 caps every launch at four workgroups, so every input
    beyond four tiles takes the multi-round form — tiles handed out through the atomic ticket (index-assigned
    tiles are only deadlock-free while every tile has its own workgroup)"""
    import os
    import la3dm_amd
    from la3dm_amd import _lib
    forced = request.param != "occupancy"
    if forced:
        os.environ["LA3DM_SCAN_RESIDENT"] = os.environ["LA3DM_RADIX_RESIDENT"] = "4"
    try:
        m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)     # only for its context
        H = _lib.hip()
        dm = C.c_void_p()
        assert H.la3dm_devmap_create(m.ctx(), C.byref(dm)) == 0
    finally:
        os.environ.pop("LA3DM_SCAN_RESIDENT", None)
        os.environ.pop("LA3DM_RADIX_RESIDENT", None)
    yield H, dm, forced
    H.la3dm_devmap_destroy(dm)


SIZES = [1, 2, 63, 64, 65, 4095, 4096, 4097, 12345, 262144, 1000003, 4096 * 512, 4096 * 512 + 1, 3_000_001, 4096 * 1024 + 17, 6_000_011, 7, 4096 * 1024, 5,
         # the sort's two-level tile prefix (groups of 16 tiles) and its two tile shapes (sixteen waves up to 256 tiles, four waves beyond)
         4096 * 16, 4096 * 16 + 1, 4096 * 17 - 1, 4096 * 33, 4096 * 256, 4096 * 256 + 1]


def test_exclusive_scan(devmap):
    H, dm, forced = devmap
    rng = np.random.default_rng(11)
    for n in SIZES:
        x = rng.integers(0, 9, n).astype(np.uint32)
        out, aux = np.zeros(n, np.uint32), np.zeros(4, np.uint32)
        assert H.la3dm_devmap_diag_scan(dm, 0, x.ctypes.data, n, out.ctypes.data, aux.ctypes.data) == 0, n
        ref = np.concatenate([[0], np.cumsum(x, dtype=np.uint64)[:-1]]).astype(np.uint32)
        assert (out == ref).all(), n
        assert int(aux[0]) == int(x.sum(dtype=np.uint64)), n


def test_head_flags_and_segment_starts(devmap):
    H, dm, forced = devmap
    rng = np.random.default_rng(12)
    for n in SIZES:
        for invalid in (0, min(n, 3), n if n < 100 else n // 5):
            nv = n - invalid
            keys = np.sort(rng.integers(0, max(2, nv // 3 + 1), nv)).astype(np.uint32)
            keys = np.concatenate([keys, np.full(invalid, 0xFFFFFFFF, np.uint32)])
            out, aux = np.zeros(n, np.uint32), np.zeros(n + 3, np.uint32)
            assert H.la3dm_devmap_diag_scan(dm, 1, keys.ctypes.data, n, out.ctypes.data, aux.ctypes.data) == 0
            flag = np.zeros(n, np.uint32)
            if nv:
                flag[0] = 1
                flag[1:nv] = keys[1:nv] != keys[:nv - 1]
            ref = np.concatenate([[0], np.cumsum(flag)[:-1]]).astype(np.uint32)
            assert (out == ref).all(), (n, invalid)
            nseg = int(flag.sum())
            assert int(aux[0]) == nseg and int(aux[1]) == nv, (n, invalid, aux[:2])
            starts = np.flatnonzero(flag).astype(np.uint32)
            assert (aux[2:2 + nseg] == starts).all() and int(aux[2 + nseg]) == nv, (n, invalid)


@pytest.mark.parametrize("bits", [1, 8, 9, 16, 22, 24, 32])
def test_stable_radix_sort(devmap, bits):
    H, dm, forced = devmap
    rng = np.random.default_rng(100 + bits)
    for n in SIZES:
        if forced and n > 1_100_000:   # (with four workgroups per launch everything beyond four tiles takes the ticket form already)
            continue
        kinds = [rng.integers(0, 2 ** bits, n, dtype=np.uint64)]
        if n > 64 and bits >= 16:
            kinds.append(rng.integers(0, 5, n, dtype=np.uint64) << (bits - 3))                  # few distinct keys, high digits only
            kinds.append(np.minimum(rng.geometric(0.01, n), 2 ** bits - 1).astype(np.uint64))    # skewed
            kinds.append(np.full(n, (2 ** bits - 1) // 3, np.uint64))                            # one key: every pass is a copy
        if n > 6_000_000 or n in (4096 * 256, 4096 * 256 + 1):
            kinds = kinds[:1]
        for keys64 in kinds:
            keys = keys64.astype(np.uint32)
            if bits == 32 and n > 10:
                keys[rng.integers(0, n, 3)] = 0xFFFFFFFF                                         # the invalid key sorts last
            vals = np.arange(n, dtype=np.uint32)
            ko, vo = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
            assert H.la3dm_devmap_diag_sort(dm, keys.ctypes.data, vals.ctypes.data, n, bits, ko.ctypes.data, vo.ctypes.data) == 0
            order = np.argsort(keys, kind="stable").astype(np.uint32)
            assert (vo == order).all(), (n, bits)
            assert (ko == keys[order]).all(), (n, bits)
