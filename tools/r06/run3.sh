#!/bin/bash
# round 6, GPU call 3: wave-per-beam grid traversal — tests, the lv leg, a 200 k-ray BGK-LV insert, kernel trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest -x -q tests/test_lv_gpu.py::test_ray_shortening_on_the_hit_grid "tests/test_baseline_configs_gpu.py::test_config3_lv_synthetic_50k_rays" > $O/run3_tests.log 2>&1
tail -3 $O/run3_tests.log
rm -rf $O/lv_trace; timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/lv_trace -o t -- python bench.py --workload lv --steps 10 --warmup 2 --no-cpu > $O/bench_lv.json 2> $O/lv_trace.log
python - <<PY
import json,csv
d=json.loads([l for l in open("$O/bench_lv.json") if l.startswith("{")][-1])
print("sequence_ms", d["leg"]["sequence_ms"], "50k insert ms", d["leg"]["synthetic_50k"]["ms_per_insert"])
rows=list(csv.DictReader(open("$O/lv_trace/t_kernel_stats.csv")))
for r in rows[:10]: print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), f'{float(r["AverageNs"])/1e3:9.1f} us avg', f'{float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
LA3DM_DEBUG_LV=1 timeout 600 python - <<PY
import time, numpy as np, la3dm_amd
params = dict(la3dm_amd.LV_YAML, resolution=0.05, block_depth=5)
xyz, origin = la3dm_amd.synthetic_scan(200000)
m = la3dm_amd.BGKLVOctoMap(**params, device=0)
ts=[]
for k in range(4):
    t0=time.perf_counter(); m.insert_pointcloud(xyz, origin, 0.05, 0.1, 8.0); ts.append((time.perf_counter()-t0)*1e3)
print("200k-ray BGK-LV inserts (ms):", [round(t,2) for t in ts], m.lv_stats())
PY
