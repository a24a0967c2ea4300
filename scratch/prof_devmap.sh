#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_devmap; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- python scratch/devmap_loop.py "$@" > $OUT/log.txt 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
for r in rows[:40]:
    print("%-70s calls %5s avg %10.1f us total %8.3f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
