#!/bin/bash
# Regenerates the profiled numbers bench.py quotes, for the kernel source of THIS build, in both accumulate modes:
#   profiles/$ROUND/bench_kernel_stats_sum{0,1}.csv  rocprofv3 --kernel-trace --stats of the bench's timed region
#   profiles/$ROUND/bench_pmc_summary.txt            counters per launch (separate --pmc passes, as the guide prescribes)
#   profiles/$ROUND/phase_table.txt                  VALU / SALU / LDS instructions per tile by phase (ablate option)
#   profiles/bgk_traffic.json                     per-launch HBM traffic + instruction counts, stamped with the kernel source hash
# usage (GPU box): bash tools/prof/update_traffic.sh      ->  results under gpurun_out/$ROUND/prof/, copy into profiles/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ROUND=${ROUND:-r06}   # output directory under gpurun_out/ and profiles/; the entries' "round" field
OUT=$GRAFT_REPO_ROOT/gpurun_out/$ROUND/prof; rm -rf $OUT; mkdir -p $OUT
for SUM in 1 0; do
  BENCH="python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e --no-big --no-side --no-other-mode --sum $SUM"
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace$SUM -o t -- $BENCH > $OUT/bench_trace$SUM.log 2>&1
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
             "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VMEM_WR" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    n=$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 --output-format csv --pmc $set -d $OUT/pmc${SUM}_$n -o p -- $BENCH > $OUT/bench_pmc${SUM}_$n.log 2>&1
  done
  # per-phase instruction counts: ablate 1 = no kernel evaluation (C), 2 = no candidate tests (B and C), 4 = no ordered fuse (D, sum 0)
  for AB in 1 2 4 5; do
    if [ $SUM = 1 ] && [ $AB -ge 4 ]; then continue; fi
    rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d $OUT/ab${SUM}_$AB -o p -- $BENCH --ablate $AB > $OUT/bench_ab${SUM}_$AB.log 2>&1
  done
done
# block_depth 4 (the reference constructor's default): kernel trace + the counters the bench's roofline.depth4 leg quotes
BENCH4="python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e --no-big --no-side --no-other-mode --sum 1 --depth 4"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace_d4 -o t -- $BENCH4 > $OUT/bench_trace_d4.log 2>&1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --output-format csv --pmc $set -d $OUT/pmcd4_$n -o p -- $BENCH4 > $OUT/bench_pmcd4_$n.log 2>&1
done
for AB in 1 2; do
  rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d $OUT/abd4_$AB -o p -- $BENCH4 --ablate $AB > $OUT/bench_abd4_$AB.log 2>&1
done
python - <<PY
import csv, glob, collections, json, os, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
out = "$OUT"
def counters(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/" + pattern + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return agg
entries = {}
with open(out + "/bench_pmc_summary.txt", "w") as fo, open(out + "/phase_table.txt", "w") as ft:
    fo.write("# python bench.py --steps 20 --warmup 3 --no-cpu --no-e2e --no-big --no-side --no-other-mode --sum S; one rocprofv3 --pmc pass per counter set; mean per launch\n")
    ft.write("# instructions per tile (= per wave) of the predict + fuse kernel, configs[1] scan, by ablation (tools/prof/update_traffic.sh)\n")
    for S in (1, 0):
        for f in glob.glob(out + f"/trace{S}/**/*kernel_stats.csv", recursive=True):
            open(out + f"/bench_kernel_stats_sum{S}.csv", "w").write(open(f).read())
        agg = counters(f"pmc{S}_*")
        for k, d in agg.items():
            for c, v in sorted(d.items()):
                line = f"sum{S} {k:60s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
                print(line); fo.write(line + "\n")
        fuse = next((d for k, d in agg.items() if "bgk_predict_fuse" in k), None)
        name = next((k for k in agg if "bgk_predict_fuse" in k), "")
        if fuse:
            mean = lambda c: sum(fuse[c]) / len(fuse[c]) if fuse.get(c) else None
            fetch_kb, write_kb = mean("FETCH_SIZE"), mean("WRITE_SIZE")
            entries[f"rays200000_d3_r0.1_sum{S}"] = {
                "kernel": name, "round": int("$ROUND"[1:]), "kernel_sha": bench.kernel_source_hash(),
                "source": "profiles/$ROUND/bench_pmc_summary.txt (tools/prof/update_traffic.sh)",
                "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
                "raw_bytes_per_launch": (fetch_kb + write_kb) * 1024 if fetch_kb and write_kb else None,
                # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): x2
                "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024 if fetch_kb and write_kb else None,
                "note": "separate rocprofv3 --pmc passes; hbm_bytes_per_launch applies the guide's gfx950 x2 FETCH_SIZE correction "
                        "(calibrated for 16 B/lane streams: an upper bound here, the alpha/beta/key loads are 4 B/lane)",
                "valu_insts_per_launch": mean("SQ_INSTS_VALU"), "salu_insts_per_launch": mean("SQ_INSTS_SALU"),
                "lds_insts_per_launch": mean("SQ_INSTS_LDS"), "valu_active_quads_per_launch": mean("SQ_ACTIVE_INST_VALU"),
                "waves_per_launch": mean("SQ_WAVES")}
            w = mean("SQ_WAVES")
            rows = [("full", mean("SQ_INSTS_VALU") / w, mean("SQ_INSTS_SALU") / w, mean("SQ_INSTS_LDS") / w, mean("SQ_ACTIVE_INST_VALU") / w)]
            for AB in (1, 2, 4, 5):
                a = counters(f"ab{S}_{AB}")
                d = next((d for k, d in a.items() if "bgk_predict_fuse" in k), None)
                if d:
                    m2 = lambda c: sum(d[c]) / len(d[c])
                    rows.append((f"ablate {AB}", m2("SQ_INSTS_VALU") / w, m2("SQ_INSTS_SALU") / w, m2("SQ_INSTS_LDS") / w, m2("SQ_ACTIVE_INST_VALU") / w))
            ft.write(f"\nbgk_sum = {S}  ({name})\n{'':12s} {'VALU':>8s} {'SALU':>8s} {'LDS':>8s} {'VALU busy (4-cycle units)':>28s}\n")
            for r in rows:
                ft.write(f"{r[0]:12s} {r[1]:8.1f} {r[2]:8.1f} {r[3]:8.1f} {r[4]:28.1f}\n")
# block_depth 4
with open(out + "/bench_pmc_summary.txt", "a") as fo, open(out + "/phase_table.txt", "a") as ft:
    for f in glob.glob(out + "/trace_d4/**/*kernel_stats.csv", recursive=True):
        open(out + "/bench_kernel_stats_depth4.csv", "w").write(open(f).read())
    agg = counters("pmcd4_*")
    fo.write("# the same with --depth 4 (block_depth 4: 512-leaf blocks, eight tiles per block)\n")
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            fo.write(f"depth4 {k:60s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
    fuse = next((d for k, d in agg.items() if "bgk_predict_fuse" in k), None)
    name = next((k for k in agg if "bgk_predict_fuse" in k), "")
    if fuse:
        mean = lambda c: sum(fuse[c]) / len(fuse[c]) if fuse.get(c) else None
        fetch_kb, write_kb = mean("FETCH_SIZE"), mean("WRITE_SIZE")
        entries["rays200000_d4_r0.1_sum1"] = {
            "kernel": name, "round": int("$ROUND"[1:]), "kernel_sha": bench.kernel_source_hash(),
            "source": "profiles/$ROUND/bench_pmc_summary.txt (tools/prof/update_traffic.sh, --depth 4)",
            "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
            "raw_bytes_per_launch": (fetch_kb + write_kb) * 1024 if fetch_kb and write_kb else None,
            "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024 if fetch_kb and write_kb else None,
            "note": "separate rocprofv3 --pmc passes; hbm_bytes_per_launch applies the guide's gfx950 x2 FETCH_SIZE correction",
            "valu_insts_per_launch": mean("SQ_INSTS_VALU"), "salu_insts_per_launch": mean("SQ_INSTS_SALU"),
            "lds_insts_per_launch": mean("SQ_INSTS_LDS"), "valu_active_quads_per_launch": mean("SQ_ACTIVE_INST_VALU"),
            "waves_per_launch": mean("SQ_WAVES")}
        w = mean("SQ_WAVES")
        rows = [("full", mean("SQ_INSTS_VALU") / w, mean("SQ_INSTS_SALU") / w, mean("SQ_INSTS_LDS") / w, mean("SQ_ACTIVE_INST_VALU") / w)]
        for AB in (1, 2):
            a = counters(f"abd4_{AB}")
            d = next((d for k, d in a.items() if "bgk_predict_fuse" in k), None)
            if d:
                m2 = lambda c: sum(d[c]) / len(d[c])
                rows.append((f"ablate {AB}", m2("SQ_INSTS_VALU") / w, m2("SQ_INSTS_SALU") / w, m2("SQ_INSTS_LDS") / w, m2("SQ_ACTIVE_INST_VALU") / w))
        ft.write(f"\nblock_depth 4, bgk_sum = 1  ({name})\n{'':12s} {'VALU':>8s} {'SALU':>8s} {'LDS':>8s} {'VALU busy (4-cycle units)':>28s}\n")
        for r in rows:
            ft.write(f"{r[0]:12s} {r[1]:8.1f} {r[2]:8.1f} {r[3]:8.1f} {r[4]:28.1f}\n")
json.dump(entries, open(out + "/bgk_traffic.json", "w"), indent=1)
print(open(out + "/phase_table.txt").read())
PY
