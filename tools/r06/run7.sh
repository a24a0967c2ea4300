#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest -x -q tests/test_lv_gpu.py::test_ray_shortening_on_the_hit_grid tests/test_lv_sum_gpu.py::test_synthetic_scan_with_split_cubes 2>&1 | tail -3
for G in 8192 2048 1024 256; do
LA3DM_LV_GRID_MIN=$G timeout 600 python bench.py --workload lv --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('grid from $G hits: sequence_ms', round(d['leg']['sequence_ms'],3), '50k insert ms', round(d['leg']['synthetic_50k']['ms_per_insert'],3))"
done
timeout 300 python tools/check/outliers.py 1500 | tail -8
