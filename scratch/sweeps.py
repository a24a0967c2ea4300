import sys, os
sys.path.insert(0, os.getcwd())
import la3dm_amd
m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
for lo in (1e-30, 1e-37):
    print("lo", lo)
    print(" div3 :", m.diag_sweep(0, lo, 3.0), m.diag_sweep(0, -lo, -3.0))
    print(" div2pi (2 steps):", m.diag_sweep(1, lo, 1.0), m.diag_sweep(1, -lo, -1.0))
    print(" div2pi (1 step):", m.diag_sweep(5, lo, 1.0), m.diag_sweep(5, -lo, -1.0))
    print(" sqrt :", m.diag_sweep(2, lo, 4.0))
print("zeros:", m.diag_sweep(0, 0.0, 0.0), m.diag_sweep(1, 0.0, 0.0), m.diag_sweep(2, 0.0, 0.0), m.diag_sweep(1, -0.0, -0.0))
