"""Minimal PCD v0.7 reader for the replay harness (the reference uses pcl::io::loadPCDFile,
src/bgkoctomap/bgkoctomap_static_node.cpp:7-16: xyz fields + sensor origin from VIEWPOINT)."""
import numpy as np


def load_pcd(path):
    """-> (xyz float32 [n,3], origin float32 [3])"""
    with open(path, "rb") as f:
        raw = f.read()
    header = {}
    pos = 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        k, _, v = line.partition(" ")
        header[k] = v.split()
        if k == "DATA":
            break
    fields = header["FIELDS"]
    sizes = [int(s) for s in header["SIZE"]]
    types = header["TYPE"]
    counts = [int(c) for c in header.get("COUNT", ["1"] * len(fields))]
    n = int(header["POINTS"][0])
    vp = [float(v) for v in header.get("VIEWPOINT", ["0", "0", "0", "1", "0", "0", "0"])]
    origin = np.asarray(vp[:3], np.float32)
    kind = header["DATA"][0]
    if kind == "binary":
        dt = []
        for name, s, t, c in zip(fields, sizes, types, counts):
            code = {"F": "f", "I": "i", "U": "u"}[t] + str(s)
            dt.append((name, code) if c == 1 else (name, code, (c,)))
        rec = np.frombuffer(raw, dtype=np.dtype(dt), count=n, offset=pos)
        xyz = np.stack([rec["x"], rec["y"], rec["z"]], axis=1).astype(np.float32)
    elif kind == "ascii":
        rows = np.loadtxt(raw[pos:].decode().splitlines()[:n], dtype=np.float64, ndmin=2)
        ix, iy, iz = fields.index("x"), fields.index("y"), fields.index("z")
        xyz = rows[:, [ix, iy, iz]].astype(np.float32)
    else:
        raise ValueError(f"unsupported PCD DATA {kind}")
    return np.ascontiguousarray(xyz), origin
