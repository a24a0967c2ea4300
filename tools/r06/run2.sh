#!/bin/bash
# round 6, GPU call 2: BGK-LV ray shortening on the hit grid — tests, the lv bench leg, its kernel trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest -x -q tests/test_lv_gpu.py tests/test_lv_sum_gpu.py "tests/test_baseline_configs_gpu.py::test_config3_lv_synthetic_50k_rays" "tests/test_baseline_configs_gpu.py::test_config3_lv_full_sequence" "tests/test_devmap_gpu.py::test_counter_block_survives_non_insert_entry_points" > $O/run2_tests.log 2>&1
tail -8 $O/run2_tests.log
LA3DM_DEBUG_LV=1 timeout 600 python bench.py --workload lv --steps 10 --warmup 2 --no-cpu > $O/bench_lv.json 2> $O/bench_lv.err; tail -c 400 $O/bench_lv.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_lv.json") if l.startswith("{")][-1])
print({k:d[k] for k in ("value","ms_per_step")}); print(json.dumps(d.get("lv",d).get("synthetic_50k", {}))[:900])
PY
rm -rf $O/lv_trace; timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/lv_trace -o t -- python bench.py --workload lv --steps 10 --warmup 2 --no-cpu > $O/lv_trace.log 2>&1
f=$(find $O/lv_trace -name '*kernel_stats.csv' | head -1); head -25 $f | cut -c1-150
