"""Phase stamps of the radix pass (library built with -DLA3DM_RS_TRACE:
    make -C la3dm_amd/csrc clean all HIPFLAGS+=-DLA3DM_RS_TRACE
la3dm_devmap_diag_sort then prints, per pass, the stamps of five tiles relative to the pass's first entry:
entry, hist read, keys loaded, ranked, published, look-back done, reordered in LDS, written).
usage: python tools/check/sort_trace.py [n] [bits] [kind: random | cells]"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
from la3dm_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 534000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 24
kind = sys.argv[3] if len(sys.argv) > 3 else "random"
rng = np.random.default_rng(1)
if kind == "cells":   # long runs of neighbouring keys, as beam samples give
    keys = (np.cumsum(rng.integers(-3, 4, n)) % (1 << bits)).astype(np.uint32)
else:
    keys = rng.integers(0, 1 << bits, n, dtype=np.uint64).astype(np.uint32)
vals = np.arange(n, dtype=np.uint32)
ko, vo = np.empty_like(keys), np.empty_like(vals)
m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
H = _lib.hip()
dm = C.c_void_p()
assert H.la3dm_devmap_create(m.ctx(), C.byref(dm)) == 0
for it in range(3):
    print("run", it, file=sys.stderr, flush=True)
    assert H.la3dm_devmap_diag_sort(dm, keys.ctypes.data, vals.ctypes.data, n, bits, ko.ctypes.data, vo.ctypes.data) == 0
order = np.argsort(keys, kind="stable").astype(np.uint32)
assert (vo == order).all()
H.la3dm_devmap_destroy(dm)
