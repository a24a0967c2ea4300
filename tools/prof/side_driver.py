"""the side legs' inserts on their own, for the profiler (tools/prof/side_pmc.sh): N identical inserts after the first
   python tools/prof/side_driver.py lv50k | lvseq | l   [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import la3dm_amd

if __name__ == "__main__":
    what = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    if what == "l":
        m = la3dm_amd.BGKLOctoMap(**dict(la3dm_amd.L_YAML, resolution=0.1, block_depth=3), device=0)
        xyz, origin = la3dm_amd.synthetic_scan(200000)
        for _ in range(n + 1):
            m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0)
    elif what == "lv50k":
        m = la3dm_amd.BGKLVOctoMap(**dict(la3dm_amd.LV_YAML, resolution=0.05, block_depth=5), device=0)
        xyz, origin = la3dm_amd.synthetic_scan(50000)
        for _ in range(n + 1):
            m.insert_pointcloud(xyz, origin, 0.05, 0.1, 8.0)
    else:
        scans = [la3dm_amd.load_pcd(os.path.join(ROOT, "tests", "golden", "data", "sim_unstructured", f"sim_unstructured_{i}.pcd"))
                 for i in range(1, 13)]
        for _ in range(n + 1):
            m = la3dm_amd.BGKLVOctoMap(**dict(la3dm_amd.LV_YAML, resolution=0.05, block_depth=5), device=0)
            for xyz, origin in scans:
                m.insert_pointcloud(xyz, origin, 0.05, 0.1, 8.0)
    print("inserts", n + 1)
