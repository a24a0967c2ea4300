#!/bin/bash
# counters of the GP training kernels in the depth-4 run (separate --pmc passes)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03/gptrain_pmc; rm -rf $OUT; mkdir -p $OUT
B="python bench.py --workload gp --depth 4 --steps 1 --warmup 1 --no-cpu"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --output-format csv --pmc $set -d $OUT/pmc$i -o p -- $B > $OUT/log$i.txt 2>&1 < /dev/null
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("$OUT/pmc_summary.txt", "w") as fo:
    for k, d in agg.items():
        if "gp_train" not in k: continue
        for c, v in sorted(d.items()):
            line = f"{k:50s} {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
            print(line); fo.write(line + "\n")
PY
