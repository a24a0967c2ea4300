#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_l; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python tools/prof/l_timing.py ${1:-200000} ${2:-2048} > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print("%-60s calls %4s avg %10.1f us max %10.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3))
PY
tail -3 $OUT/log.txt
