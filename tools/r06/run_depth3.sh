#!/bin/bash
# round 6, second session: the block_depth-3 leaf-list / write-back launches with several test blocks per wave (devmap_depth3.h)
# — parity first (the suites that meet pruned pools, sharded ranges and the A/B switch), then A/B timings and the timelines.
#   gpurun --timeout 900 -- 'bash tools/r06/run_depth3.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/depth3; mkdir -p $O
timeout 420 python -m pytest -q -m gpu -x tests/test_devmap_gpu.py tests/test_sharded_insert_gpu.py "tests/test_bgk_gpu.py" \
   "tests/test_baseline_configs_gpu.py::test_config1_bgk_200k_rays" tests/test_bgk_sum_gpu.py::test_config4_scan_1m_rays 2>&1 | tail -6 | tee $O/tests.txt
for v in 0 1; do
  LA3DM_DEPTH3=$v timeout 200 python bench.py --gpus 1 --mode shard --steps 20 --warmup 3 > $O/shard_$v.json 2> $O/shard_$v.err
  LA3DM_DEPTH3=$v timeout 200 python bench.py --no-side --no-cpu --no-big --no-other-mode --steps 10 > $O/e2e_$v.json 2> $O/e2e_$v.err
done
python - <<PY
import json
for v in (0, 1):
    s = json.loads([l for l in open("$O/shard_%d.json" % v) if l.startswith("{")][-1])
    e = json.loads([l for l in open("$O/e2e_%d.json" % v) if l.startswith("{")][-1])
    print("LA3DM_DEPTH3=%d  configs[4] insert %.4f ms   configs[1] insert %.4f ms (cloud in HBM) %.4f ms (with upload)  scale_n1 %.4f" % (
        v, s["ms_per_step"], e["end_to_end"]["ms_per_insert_device_cloud"], e["end_to_end"]["ms_per_insert"], e.get("scale_n1", {}).get("ms_per_step", float("nan"))))
PY
for v in 0 1; do
  LA3DM_DEPTH3=$v bash tools/prof/prof_devmap.sh 1000000 4 0.05 > /dev/null 2>&1; cp gpurun_out/prof_devmap/timeline.txt $O/timeline_1M_$v.txt
  LA3DM_DEPTH3=$v bash tools/prof/prof_devmap.sh 200000 5 0.1 > /dev/null 2>&1; cp gpurun_out/prof_devmap/timeline.txt $O/timeline_200k_$v.txt
done
for f in $O/timeline_*; do echo $f; grep -E "dm_leaves|dm_commit_prune" $f | tail -3 | cut -c1-110; done
