#!/bin/bash
# A/B of the scan outputs nobody read (devmap_grid_keys.h): LA3DM_GRID_KEYS = 0 | 1 — parity first, then the timelines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/grid_keys; mkdir -p $O
timeout 400 python -m pytest -q -m gpu -x tests/test_devmap_gpu.py tests/test_front_end_primitives_gpu.py tests/test_bgkl_gpu.py::test_host_orchestrated_mode \
   tests/test_baseline_configs_gpu.py::test_config3_lv_full_sequence "tests/test_baseline_configs_gpu.py::test_config1_bgk_200k_rays" \
   tests/test_likely_trig_gpu.py 2>&1 | tail -4 | tee $O/tests.txt
for v in 0 1; do
  LA3DM_GRID_KEYS=$v bash tools/prof/prof_devmap.sh 1000000 4 0.05 > /dev/null 2>&1; cp gpurun_out/prof_devmap/timeline.txt $O/timeline_1M_$v.txt
  LA3DM_GRID_KEYS=$v bash tools/prof/prof_devmap.sh 200000 5 0.1 > /dev/null 2>&1; cp gpurun_out/prof_devmap/timeline.txt $O/timeline_200k_$v.txt
  LA3DM_GRID_KEYS=$v timeout 200 python bench.py --gpus 1 --mode shard --steps 20 --warmup 3 > $O/shard_$v.json 2> $O/shard_$v.err
done
python - <<PY
import json
for v in (0, 1):
    s = json.loads([l for l in open("$O/shard_%d.json" % v) if l.startswith("{")][-1])
    print("LA3DM_GRID_KEYS=%d  configs[4] insert %.4f ms" % (v, s["ms_per_step"]))
PY
for f in $O/timeline_*; do echo $f; grep -E "dm_scan_lb<true>|dm_grid_centroids" $f | tail -7 | cut -c1-100; done
