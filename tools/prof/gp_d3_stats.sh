#!/bin/bash
# kernel-trace stats of the depth-3 GP run (configs[2])
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03/gp_d3; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --workload gp --steps 5 --warmup 2 --no-cpu > $OUT/log.txt 2>&1 < /dev/null
cut -d, -f1-4 $OUT/trace/t_kernel_stats.csv | head -12
