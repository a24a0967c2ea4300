#!/bin/bash
# round 6, end of the last session: the whole GPU suite with durations, the default bench line, the insert timelines
#   gpurun --timeout 1100 -- 'bash tools/r06/run_final.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/final; mkdir -p $O
timeout 800 python -m pytest tests -q -m gpu --durations=15 > $O/full_gpu_suite.log 2>&1
tail -20 $O/full_gpu_suite.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/prof/prof_devmap.sh 200000 5 0.1 > /dev/null 2>&1; cp gpurun_out/prof_devmap/timeline.txt $O/devmap_timeline_200k.txt
bash tools/prof/prof_devmap.sh 1000000 4 0.05 > /dev/null 2>&1; cp gpurun_out/prof_devmap/timeline.txt $O/devmap_timeline_1M.txt
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
r=d["roofline"]
print("value", d["value"], "ms_per_step", d["ms_per_step"], "kernel_ms", r["kernel_ms"], "frac", r["frac"], "traffic", r.get("traffic"))
print("scale_n1", d.get("scale_n1",{}).get("ms_per_step"), "e2e", d.get("end_to_end",{}).get("ms_per_insert"), d.get("end_to_end",{}).get("ms_per_insert_device_cloud"))
print("gp", d["gp"]["depth3"]["ms_per_step"], d["gp"]["depth4"]["ms_per_step"], "lv", d["lv"]["sequence_ms"], d["lv"]["synthetic_50k"]["ms_per_insert"], "bgkl", d["bgkl"]["ms_per_step"])
PY
head -1 $O/devmap_timeline_200k.txt $O/devmap_timeline_1M.txt
