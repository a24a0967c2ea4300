import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import la3dm_amd
m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
print("sqrt [0,0]:", m.diag_sweep(2, 0.0, 0.0), " (0, 1e-30]:", m.diag_sweep(2, 1e-45, 1e-30), " [1e-30, 4]:", m.diag_sweep(2, 1e-30, 4.0))
edges = [0.0, 1e-30, 0.5, 0.785, 0.786, 1.0, 2.0, 2.3, 2.4, 3.0, 3.9, 4.0, 5.0, 5.4, 5.6, 6.0, 6.2831855]
for a, b in zip(edges[:-1], edges[1:]):
    print(f"sincos [{a}, {b}]: {m.diag_sweep(3, a, b)}")
