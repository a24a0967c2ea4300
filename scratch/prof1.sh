#!/bin/bash
# usage: prof1.sh <tag> <bench args...>

TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --steps 20 --warmup 3 --no-cpu "$@" > $OUT/bench_trace.log 2>&1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --output-format csv --pmc $set -d $OUT/pmc_$n -o p -- python bench.py --steps 3 --warmup 1 --no-cpu "$@" > $OUT/bench_pmc_$n.log 2>&1
done
find $OUT -name "*.csv" | head -40
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:2000])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/pmc_summary.txt", "w") as fo:
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            line = f"{k:60s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
            print(line); fo.write(line + "\n")
PY
