#!/bin/bash
# counters of the depth-3 GP kernels (configs[2]); separate --pmc passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03/gp_d3_pmc; rm -rf $OUT; mkdir -p $OUT
B="python bench.py --workload gp --steps 1 --warmup 1 --no-cpu"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" ; do
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
        if "gp_" not in k: continue
        for c, v in sorted(d.items()):
            line = f"{k:50s} {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
            print(line); fo.write(line + "\n")
PY
