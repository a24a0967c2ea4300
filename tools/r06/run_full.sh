#!/bin/bash
# round 6: the whole GPU suite with durations, then (RESTAMP=1) every stamped counter file of this tree (tools/prof/restamp_all.sh),
# then the default bench line and the kernel-function variants
#   gpurun --timeout 4800 -- 'RESTAMP=1 bash tools/r06/run_full.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=25 > $O/full_gpu_suite.log 2>&1
tail -34 $O/full_gpu_suite.log
export ROUND=r06
if [ "$RESTAMP" = "1" ]; then
  timeout 2400 bash tools/prof/restamp_all.sh > $O/restamp.log 2>&1
  tail -4 $O/restamp.log
fi
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.err
timeout 600 python bench.py --variants --no-side --no-cpu --no-e2e > $O/bench_variants.json 2> $O/bench_variants.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
r=d["roofline"]
print("value", d["value"], "ms_per_step", d["ms_per_step"], "kernel_ms", r["kernel_ms"], "frac", r["frac"], "traffic", r.get("traffic"))
print("depth4", {k: r["depth4"].get(k) for k in ("kernel_ms","frac","traffic")}, r["depth4"].get("issue"))
print("ooc", r["out_of_cache"]["kernel_ms"], r["out_of_cache"]["frac"])
print("scale_n1", d.get("scale_n1",{}).get("ms_per_step"), "e2e", d.get("end_to_end",{}).get("ms_per_insert"), d.get("end_to_end",{}).get("ms_per_insert_device_cloud"))
print("gp", d["gp"]["depth3"]["ms_per_step"], d["gp"]["depth3"].get("gp_mode_1",{}).get("ms_per_step"), d["gp"]["depth4"]["ms_per_step"])
print("lv", d["lv"]["sequence_ms"], d["lv"]["synthetic_50k"]["ms_per_insert"], "bgkl", {k:v for k,v in d["bgkl"].items() if k.startswith("ms_")})
print("cpu", d["cpu_baseline"]["value"], d.get("cpu_baseline_omp",{}).get("value"), d.get("cpu_baseline_omp",{}).get("cores"))
v=json.loads([l for l in open("$O/bench_variants.json") if l.startswith("{")][-1])["roofline"]["kernel_function_variants"]
for k,x in v.items(): print(k, round(x["configs1_kernel_us"],1), round(x["configs1_frac"],3), round(x["out_of_cache_kernel_us"],0), round(x["out_of_cache_frac"],3))
PY
