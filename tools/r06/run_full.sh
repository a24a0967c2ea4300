#!/bin/bash
# round 6: the whole GPU suite with durations, then every stamped counter file of this tree (tools/prof/restamp_all.sh), then the default bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=40 > $O/full_gpu_suite.log 2>&1
tail -60 $O/full_gpu_suite.log
export ROUND=r06
timeout 2400 bash tools/prof/restamp_all.sh > $O/restamp.log 2>&1
tail -12 $O/restamp.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
r=d["roofline"]
print("value", d["value"], "ms_per_step", d["ms_per_step"], "kernel_ms", r["kernel_ms"], "frac", r["frac"], "traffic", r.get("traffic"))
print("depth4", {k: r["depth4"].get(k) for k in ("kernel_ms","frac","issue")})
print("scale_n1", d.get("scale_n1",{}).get("ms_per_step"), "e2e", d.get("end_to_end",{}).get("ms_per_insert"))
print("gp", d["gp"]["depth3"]["ms_per_step"], d["gp"]["depth3"].get("gp_mode_1",{}).get("ms_per_step"), d["gp"]["depth4"]["ms_per_step"])
print("lv", d["lv"]["sequence_ms"], d["lv"]["synthetic_50k"]["ms_per_insert"], "bgkl", d["bgkl"]["ms_per_step"])
PY
