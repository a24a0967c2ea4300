#!/bin/bash
# kernel stats of BGK-L inserts in the library's default mode: tools/prof/prof_l_loop.sh [rays] [inserts]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_l_loop; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python tools/prof/l_loop.py ${1:-200000} ${2:-5} > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
n = ${2:-5}
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-64s calls/insert %5.1f  us/insert %8.1f" % (r["Name"][:64], int(r["Calls"]) / n, float(r["TotalDurationNs"]) / n / 1e3))
PY
tail -2 $OUT/log.txt
