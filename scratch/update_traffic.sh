#!/bin/bash
# Regenerates the profiled numbers bench.py quotes, for the kernel source of THIS build:
#   profiles/r02/bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench's timed region
#   profiles/r02/bench_pmc_summary.txt    counters per launch (separate --pmc passes, as the guide prescribes)
#   profiles/bgk_traffic.json             per-launch HBM traffic + instruction counts, stamped with the kernel source hash
# usage (GPU box): bash scratch/update_traffic.sh          ->  results under gpurun_out/r02/prof/, copy into profiles/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/prof; rm -rf $OUT; mkdir -p $OUT
BENCH="python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e --no-big"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- $BENCH > $OUT/bench_trace.log 2>&1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --output-format csv --pmc $set -d $OUT/pmc_$n -o p -- $BENCH > $OUT/bench_pmc_$n.log 2>&1
done
python - <<PY
import csv, glob, collections, json, os, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
out = "$OUT"
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    open(out + "/bench_kernel_stats.csv", "w").write(open(f).read())
    print(open(f).read()[:1500])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/bench_pmc_summary.txt", "w") as fo:
    fo.write("# python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e --no-big; one rocprofv3 --pmc pass per counter set; mean per launch\n")
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            line = f"{k:60s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
            print(line); fo.write(line + "\n")
fuse = next((d for k, d in agg.items() if "bgk_predict_fuse" in k), None)
if fuse:
    import bench
    mean = lambda c: sum(fuse[c]) / len(fuse[c]) if fuse.get(c) else None
    fetch_kb, write_kb = mean("FETCH_SIZE"), mean("WRITE_SIZE")
    entry = {"kernel": "bgk_predict_fuse_v5<0,1>", "round": 2, "kernel_sha": bench.kernel_source_hash(),  # bgk_kernels.h
             "source": "profiles/r02/bench_pmc_summary.txt (scratch/update_traffic.sh)",
             "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
             "raw_bytes_per_launch": (fetch_kb + write_kb) * 1024 if fetch_kb and write_kb else None,
             # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): x2
             "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024 if fetch_kb and write_kb else None,
             "note": "separate rocprofv3 --pmc passes; hbm_bytes_per_launch applies the guide's gfx950 x2 FETCH_SIZE correction "
                     "(calibrated for 16 B/lane streams: an upper bound here, the alpha/beta/key loads are 4 B/lane)",
             "valu_insts_per_launch": mean("SQ_INSTS_VALU"), "salu_insts_per_launch": mean("SQ_INSTS_SALU"),
             "lds_insts_per_launch": mean("SQ_INSTS_LDS"), "waves_per_launch": mean("SQ_WAVES")}
    json.dump({"rays200000_d3_r0.1": entry}, open(out + "/bgk_traffic.json", "w"), indent=1)
    print(json.dumps(entry, indent=1))
PY
