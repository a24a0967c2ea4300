#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03/prof_lv; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- python tools/prof/lv_seq.py > $OUT/log.txt 2>&1
head -25 $OUT/t_kernel_stats.csv | cut -c1-150
