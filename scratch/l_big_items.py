import sys, os, zlib
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
xyz, origin = la3dm_amd.synthetic_scan(200000)
out = []
for rows in (2048, 0):
    m = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0)
    m.set_option("bgkl_split_rows", rows)
    m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0)
    lv = m.leaves()
    out.append((lv["A"].size, zlib.crc32(lv["A"].tobytes()), zlib.crc32(lv["B"].tobytes()), zlib.crc32(lv["state"].tobytes())))
    print(rows, out[-1], flush=True)
print("SAME" if out[0] == out[1] else "DIFFERENT")
