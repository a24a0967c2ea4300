#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_gp; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --workload gp --depth 4 --steps 3 --warmup 1 --no-cpu > $OUT/log.txt 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT/pmc -o p -- python bench.py --workload gp --depth 4 --steps 1 --warmup 1 --no-cpu > $OUT/log_pmc.txt 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS -d $OUT/pmc2 -o p -- python bench.py --workload gp --depth 4 --steps 1 --warmup 1 --no-cpu > $OUT/log_pmc2.txt 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 -d $OUT/pmc3 -o p -- python bench.py --workload gp --depth 4 --steps 1 --warmup 1 --no-cpu > $OUT/log_pmc3.txt 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("%-60s calls %4s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
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
tail -2 $OUT/log_pmc.txt | cut -c1-300
