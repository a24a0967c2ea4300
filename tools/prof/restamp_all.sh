#!/bin/bash
# Regenerates EVERY hash-stamped counter file bench.py quotes, for the kernel sources of this tree, and leaves them (with
# the per-round summaries) under gpurun_out/$ROUND/stamped/ — copy into profiles/ and commit as the LAST commit that touches
# a kernel header (tests/test_profiles_stamps_cpu.py is red until then).
#   gpurun --timeout 2400 -- 'bash tools/prof/restamp_all.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export ROUND=${ROUND:-r06}
S=$GRAFT_REPO_ROOT/gpurun_out/$ROUND/stamped; rm -rf $S; mkdir -p $S/$ROUND
bash tools/prof/update_traffic.sh > $S/log_update_traffic.txt 2>&1
cp gpurun_out/$ROUND/prof/bgk_traffic.json $S/ && cp gpurun_out/$ROUND/prof/{bench_pmc_summary.txt,phase_table.txt,bench_kernel_stats_sum0.csv,bench_kernel_stats_sum1.csv,bench_kernel_stats_depth4.csv} $S/$ROUND/
bash tools/prof/gp_counters.sh > $S/log_gp_counters.txt 2>&1
cp gpurun_out/$ROUND/gp_counters/gp_counters.json $S/
bash tools/prof/side_pmc.sh > $S/log_side_pmc.txt 2>&1
cp gpurun_out/$ROUND/side/side_counters.json $S/ && cp gpurun_out/$ROUND/side/side_pmc_*.txt $S/$ROUND/
for LEG in lv50k lvseq l; do f=$(find gpurun_out/$ROUND/side/${LEG}_trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $S/$ROUND/side_${LEG}_kernel_stats.csv; done
ls -la $S $S/$ROUND
python -m pytest tests/test_profiles_stamps_cpu.py -q 2>&1 | tail -3   # (still red here: profiles/ holds the old files until they are copied)
