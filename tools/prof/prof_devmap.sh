#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_devmap; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- python tools/prof/devmap_loop.py "$@" > $OUT/log.txt 2>&1
python tools/prof/timeline.py $OUT/t_kernel_trace.csv > $OUT/timeline.txt
head -3 $OUT/timeline.txt; tail -5 $OUT/log.txt
