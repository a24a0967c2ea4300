#!/bin/bash
# kernel-trace stats of the depth-4 GP run, once per value of LA3DM_GP_DBG given on the command line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03/gptrain; mkdir -p $OUT
for D in "$@"; do
  LA3DM_GP_DBG=$D timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace$D -o t -- python bench.py --workload gp --depth 4 --steps 3 --warmup 1 --no-cpu > $OUT/log$D.txt 2>&1 < /dev/null
  echo "dbg $D"; grep gp_train $OUT/trace$D/t_kernel_stats.csv | cut -d, -f1-4
done
