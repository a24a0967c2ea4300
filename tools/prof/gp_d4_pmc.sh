#!/bin/bash
# counters of the depth-4 GP predict + fuse kernel (matrix-core path): where its time goes besides the MFMAs
# usage (GPU box): bash tools/prof/gp_d4_pmc.sh -> gpurun_out/r04/gp_d4_pmc.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04/gp_d4; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --output-format csv --pmc $set -d $OUT/p$i -o p -- python bench.py --workload gp --depth 4 --steps 1 --warmup 1 --no-cpu > $OUT/log$i.txt 2>&1 < /dev/null
done
python - <<PY > $GRAFT_REPO_ROOT/gpurun_out/r04/gp_d4_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gp_" in row["Kernel_Name"]:
            agg[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# python bench.py --workload gp --depth 4 --steps 1 --warmup 1 --no-cpu under rocprofv3 --pmc (three passes); mean per launch")
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("    %-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/r04/gp_d4_pmc.txt
