#!/bin/bash
# usage: pmc_quick.sh <bench args...>   -> instruction counts per wave of the predict kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=/tmp/pmcq; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY -d $OUT -o p -- python bench.py --steps 3 --warmup 1 --no-cpu --no-e2e "$@" > $OUT/log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "fuse" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
w = sum(agg["SQ_WAVES"]) / len(agg["SQ_WAVES"])
print("$*", {k: round(sum(v) / len(v) / w, 1) for k, v in agg.items() if k != "SQ_WAVES"})
PY
