#!/bin/bash
# round 6, GPU call 4: fused per-beam LV kernel; GP mode 1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest -x -q tests/test_lv_gpu.py::test_ray_shortening_on_the_hit_grid "tests/test_baseline_configs_gpu.py::test_config3_lv_synthetic_50k_rays" > $O/run4_tests_lv.log 2>&1
tail -3 $O/run4_tests_lv.log
timeout 1200 python -m pytest -x -q tests/test_gp_gpu.py > $O/run4_tests_gp.log 2>&1
tail -15 $O/run4_tests_gp.log
rm -rf $O/lv_trace; timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/lv_trace -o t -- python bench.py --workload lv --steps 10 --warmup 2 --no-cpu > $O/bench_lv.json 2> $O/lv_trace.log
python - <<PY
import json,csv
d=json.loads([l for l in open("$O/bench_lv.json") if l.startswith("{")][-1])
print("sequence_ms", d["leg"]["sequence_ms"], "50k insert ms", d["leg"]["synthetic_50k"]["ms_per_insert"])
rows=list(csv.DictReader(open("$O/lv_trace/t_kernel_stats.csv")))
for r in rows[:10]: print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), f'{float(r["AverageNs"])/1e3:9.1f} us avg', f'{float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
timeout 600 python bench.py --workload lv --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('untraced: sequence_ms', d['leg']['sequence_ms'], '50k insert ms', d['leg']['synthetic_50k']['ms_per_insert'])"
timeout 600 python bench.py --workload gp --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); l=d['leg']; print('gp d3: mode0 ms', l['ms_per_step'], 'kernel', l['roofline']['kernel_ms'], 'mode1', l.get('gp_mode_1'))"
