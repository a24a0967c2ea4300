#!/bin/bash
# round 6, GPU call 5: per-tile descriptors at depth 4, grid_order, LV stress case; depth-4 timing with / without
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest -x -q tests/test_bgk_sum_gpu.py::test_per_tile_descriptors_change_nothing tests/test_likely_trig_gpu.py::test_likely_reference_build_end_to_end_bit_identical "tests/test_lv_gpu.py::test_ray_shortening_on_the_hit_grid" "tests/test_bgk_sum_gpu.py::test_config1_bgk_200k_rays" "tests/test_sharded_insert_gpu.py::test_sharded_replicas_equal_the_single_process_map" > $O/run5_tests.log 2>&1
tail -15 $O/run5_tests.log
timeout 900 python - <<PY
import sys, json
sys.path.insert(0, ".")
import torch, bench, la3dm_amd
from la3dm_amd import _lib
dev = torch.device("cuda", 0)
for name, opts in (("tile desc (default)", None), ("block desc", {"bgk_tile_desc": 0}), ("tile desc, ablate 1 (no evaluation)", {"ablate": 1}), ("tile desc, ablate 2 (no tests, no evaluation)", {"ablate": 2}),
                   ("block desc, ablate 2", {"bgk_tile_desc": 0, "ablate": 2})):
    r = bench.packed_kernel_leg(la3dm_amd, _lib, torch, dev, 200000, 0.1, 4, "d4", steps=20, options=opts)
    print(f"depth 4 {name:48s} kernel {r['kernel_ms']*1e3:8.1f} us  step {r['ms_per_step']*1e3:8.1f} us  frac {r['frac']:.3f}  {r['kernel']}")
PY
