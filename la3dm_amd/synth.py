"""Synthetic scans of BASELINE.md §4 / SURVEY.md §8(d): a room with 24 spheres scanned by
n rays on a Fibonacci sphere from (0,0,1), nearest hit + N(0, 0.01 m) range noise."""
import numpy as np


def synthetic_scan(n_rays, seed=1234, noise=0.01, origin=None):
    """-> (xyz float32 [n,3], origin float32 [3]).  `origin` moves the sensor inside the same room (the spheres only
    depend on `seed` and keep clear of the default pose (0, 0, 1)); the default call is the BASELINE.md §4 scan."""
    rng = np.random.default_rng(seed)
    sensor = np.array([0.0, 0.0, 1.0]) if origin is None else np.asarray(origin, np.float64)
    origin = np.array([0.0, 0.0, 1.0])
    centres, radii = [], []
    while len(centres) < 24:
        c = np.array([rng.uniform(-9, 9), rng.uniform(-9, 9), rng.uniform(0.3, 2.0)])
        r = rng.uniform(0.3, 1.0)
        if np.linalg.norm(c - origin) < r + 0.5:
            continue
        centres.append(c)
        radii.append(r)
    centres = np.asarray(centres)
    radii = np.asarray(radii)
    origin = sensor
    if (np.linalg.norm(centres - origin, axis=1) < radii + 0.05).any():
        raise ValueError("synthetic_scan: the sensor pose lies inside a sphere of the scene")
    i = np.arange(n_rays, dtype=np.float64) + 0.5
    z = 1.0 - 2.0 * i / n_rays
    phi = np.pi * (1.0 + 5.0 ** 0.5) * i
    s = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    d = np.stack([s * np.cos(phi), s * np.sin(phi), z], axis=1)
    # room [-10,10]x[-10,10]x[0,5]: nearest positive slab exit
    lo = np.array([-10.0, -10.0, 0.0])
    hi = np.array([10.0, 10.0, 5.0])
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (lo - origin) / d
        t2 = (hi - origin) / d
    t = np.where(d > 0, t2, np.where(d < 0, t1, np.inf)).min(axis=1)
    for c, r in zip(centres, radii):
        oc = origin - c
        b = d @ oc
        disc = b * b - (oc @ oc - r * r)
        ok = disc > 0
        ts = -b - np.sqrt(np.where(ok, disc, 0.0))
        ok &= ts > 0
        t = np.where(ok & (ts < t), ts, t)
    t = t + rng.normal(0.0, noise, size=n_rays)
    xyz = origin + d * t[:, None]
    return np.ascontiguousarray(xyz, np.float32), origin.astype(np.float32)
