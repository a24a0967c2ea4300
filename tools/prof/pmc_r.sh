#!/bin/bash
# usage: pmc_r.sh <bench args...>  -> per-wave counters of the predict kernel (three --pmc passes)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=/tmp/pmcr; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $set -d $OUT/p$i -o p -- python bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --no-big --no-side --no-other-mode "$@" > $OUT/log$i 2>&1 || tail -3 $OUT/log$i
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bgk_predict_fuse" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
w = sum(agg["SQ_WAVES"]) / len(agg["SQ_WAVES"])
print("$*  waves", w)
for k, v in sorted(agg.items()):
    if k != "SQ_WAVES": print(f"  {k:28s} per wave {sum(v) / len(v) / w:10.1f}")
PY
