import numpy as np, sys
sys.path.insert(0, '.')
import la3dm_amd
rng = np.random.default_rng(2026)
for case in range(12):
    res = float(rng.choice([0.05, 0.1, 0.2]))
    depth = int(rng.choice([1, 2, 3, 4]))
    params = dict(resolution=res, block_depth=depth, sf2=float(rng.choice([0.1, 1.0])),
                  ell=float(rng.choice([1.5, 2.0, 3.0])) * res, free_thresh=0.3, occupied_thresh=0.7,
                  var_thresh=float(rng.choice([0.05, 100.0])), prior_A=0.001, prior_B=0.001)
    md = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(True)
    mh = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
    for scan in range(3):
        n = int(rng.integers(1, 400))
        origin = rng.uniform(-1, 1, 3).astype(np.float32)
        pts = origin + rng.normal(0, 1.0, (n, 3)).astype(np.float32) * rng.uniform(0.2, 3.0)
        k = n // 4
        pts[:k] = np.round(pts[:k] / res) * res
        ds = float(rng.choice([-1.0, res, 2 * res]))
        fr = float(rng.choice([0.3, 0.5, 1.0])) * max(res * 4, 0.2)
        mr = float(rng.choice([-1.0, 2.5, 6.0]))
        for name, m in (("md", md), ("mh", mh)):
            try:
                m.insert_pointcloud(pts, origin, ds, fr, mr)
            except RuntimeError as e:
                print("FAIL", case, scan, name, n, ds, fr, mr, params, str(e)[-80:])
                sys.exit(1)
print("all ok")
