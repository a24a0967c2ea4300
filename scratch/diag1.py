import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
from oracle import oracle as O
m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
rng = np.random.default_rng(0)
r = np.concatenate([np.linspace(0, 1.0, 4000001), rng.uniform(0.9, 1.0, 1000000), rng.uniform(0.0, 1.0, 3000000)]).astype(np.float32)
k_or = O.kernel(r).astype(np.float64)
for op in (3, 8, 11):
    k = m.diag_eval(op, r).astype(np.float64)
    d = np.abs(k - k_or)
    print("kernel op", op, "max abs err", d.max(), "rim(r>0.9)", d[r > 0.9].max(), "mismatch frac", (d != 0).mean())
t = (r * np.float32(2.0)) * np.float32(3.1415926)
for op, fn in ((9, np.sin), (10, np.cos), (6, np.sin), (7, np.cos)):
    g = m.diag_eval(op, t)
    ref = fn(t.astype(np.float64)).astype(np.float32)
    print("trig op", op, "mismatch vs correctly rounded:", (g != ref).mean(), "max abs", np.abs(g.astype(np.float64)-ref).max())
