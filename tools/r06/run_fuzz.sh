#!/bin/bash
# round 6, final build: differential fuzz against the restatement (tests/manual/fuzz_pool.py) in chunks, for BUDGET seconds —
# default accumulate mode, ordered mode, the likely-reference configuration (fast_trig 3 + grid_order 1 + gp_mode 1 against
# set_modes(1, 1) / set_gp_mode(1)), degenerate inputs, large GP blocks, large clouds.  One line per chunk in gpurun_out/r06/fuzz.log.
#   gpurun --timeout 900 -- 'BUDGET=600 bash tools/r06/run_fuzz.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
LOG=$O/fuzz.log; : > $LOG
BUDGET=${BUDGET:-600}
T0=$(date +%s)
left() { echo $(( BUDGET - ($(date +%s) - T0) )); }
run() {   # run <label> <first> <count> <flavour> [env...]
  local label=$1 first=$2 count=$3 flav=$4; shift 4
  local l=$(left); [ $l -lt 20 ] && return 1
  local out; out=$(env "$@" timeout $l python tests/manual/fuzz_pool.py $first $count $flav 2>&1 | grep -E "MISMATCH|mismatching|Error|error" | tail -3)
  echo "$label: ${out:-cut by the time budget}" | tee -a $LOG
}
S=${SEED0:-60000}
i=0
while [ $(left) -gt 40 ]; do
  a=$((S + 1000 * i))
  run "default mode       " $a 40 ""           X=1            || break
  run "ordered mode       " $((a + 100)) 40 "" LA3DM_BGK_SUM=0 || break
  run "likely reference   " $((a + 200)) 40 likely X=1        || break
  run "degenerate inputs  " $((a + 300)) 20 degenerate X=1    || break
  run "large GP blocks    " $((a + 400)) 6 gp X=1             || break
  run "large clouds       " $((a + 500)) 2 big X=1            || break
  i=$((i + 1))
done
echo "elapsed $(( $(date +%s) - T0 )) s" | tee -a $LOG
