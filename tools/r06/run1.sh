#!/bin/bash
# round 6, first GPU call: the new / changed tests, the kernel-function variants table, the depth-4 leg
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest -x -q tests/test_sharded_insert_gpu.py tests/test_abi_errors_gpu.py "tests/test_devmap_gpu.py::test_counter_block_survives_non_insert_entry_points" tests/test_bench_gpu.py::test_shard_workload_at_world_1 "tests/test_bench_gpu.py::test_two_ranks_self_test" > $O/run1_tests.log 2>&1
tail -5 $O/run1_tests.log
timeout 60 python tests/helpers/rccl_single_rank.py > $O/run1_rccl.log 2>&1; tail -3 $O/run1_rccl.log
timeout 600 python bench.py --variants --no-side --no-cpu --no-e2e > $O/bench_variants.json 2> $O/bench_variants.err; tail -c 600 $O/bench_variants.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_variants.json") if l.startswith("{")][-1])
r=d["roofline"]
print("headline", r["kernel_ms"], r["frac"], "ooc", r["out_of_cache"]["kernel_ms"], r["out_of_cache"]["frac"])
print("depth4", json.dumps(r["depth4"]))
print("scale_n1", json.dumps(d.get("scale_n1")))
for k,v in r["kernel_function_variants"].items(): print(k, v)
PY
