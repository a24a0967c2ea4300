#!/bin/bash
# PMC counters of the BGK-LV and BGK-L inserts per kernel (separate rocprofv3 --pmc passes, no trace domains beside them),
# stamped with the hashes of the kernel sources:
#   gpurun_out/$ROUND/side/side_counters.json  -> copy to profiles/side_counters.json (bench.py's lv / bgkl legs quote its traffic)
#   gpurun_out/$ROUND/side/side_pmc_<leg>.txt  -> profiles/$ROUND/
# usage (GPU box): bash tools/prof/side_pmc.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ROUND=${ROUND:-r06}   # output directory under gpurun_out/ and profiles/; the entries' "round" field
OUT=$GRAFT_REPO_ROOT/gpurun_out/$ROUND/side; rm -rf $OUT; mkdir -p $OUT
N=3
for LEG in lv50k lvseq l; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
             "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VMEM_WR" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --output-format csv --pmc $set -d $OUT/${LEG}_p$i -o p -- python tools/prof/side_driver.py $LEG $N > $OUT/log_${LEG}_$i.txt 2>&1 < /dev/null || tail -3 $OUT/log_${LEG}_$i.txt
  done
  timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${LEG}_trace -o t -- python tools/prof/side_driver.py $LEG $N > $OUT/log_${LEG}_trace.txt 2>&1 < /dev/null
done
python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
N = $N + 1
out = {}
for leg, srcs in (("lv50k", ("lv_kernels.h", "devmap_lv_kernels.h")), ("lvseq", ("lv_kernels.h", "devmap_lv_kernels.h")), ("l", ("bgkl_kernels.h",))):
    per = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> values in dispatch order
    for f in sorted(glob.glob("$OUT/%s_p*/**/*counter_collection.csv" % leg, recursive=True)):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            per[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    scans = N * (12 if leg == "lvseq" else 1)
    kern = {}
    for k, d in per.items():
        e = {}
        for c, v in d.items():
            n = len(v) // N if len(v) >= N else len(v)          # dispatches of the LAST insert (sequence) only
            e[c] = sum(v[len(v) - n:]) if n else 0.0
            e["dispatches"] = n
        kern[k] = e
    tot = collections.defaultdict(float)
    for e in kern.values():
        for c, v in e.items():
            tot[c] += v
    fetch, write = tot.get("FETCH_SIZE", 0.0), tot.get("WRITE_SIZE", 0.0)
    out[leg] = {"round": int("$ROUND"[1:]), "kernel_sha": bench.kernel_source_hash(srcs), "sources": list(srcs),
                "unit": "per insert_pointcloud" if leg != "lvseq" else "per 12-scan sequence",
                "source": "profiles/$ROUND/side_pmc_%s.txt (tools/prof/side_pmc.sh: separate rocprofv3 --pmc passes)" % leg,
                "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
                # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): x2
                "hbm_bytes": (2 * fetch + write) * 1024, "raw_bytes": (fetch + write) * 1024,
                "valu_insts": tot.get("SQ_INSTS_VALU"), "salu_insts": tot.get("SQ_INSTS_SALU"), "lds_insts": tot.get("SQ_INSTS_LDS"),
                "valu_active_units": tot.get("SQ_ACTIVE_INST_VALU"), "waves": tot.get("SQ_WAVES"),
                "kernels": {k: {c: e.get(c) for c in ("dispatches", "FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
                                                      "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
                                                      "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
                                                      "SQ_INSTS_VALU_FMA_F64", "GRBM_GUI_ACTIVE") if c in e}
                            for k, e in sorted(kern.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))}}
    with open("$OUT/side_pmc_%s.txt" % leg, "w") as fo:
        fo.write("# python tools/prof/side_driver.py %s %d under rocprofv3 --pmc (one pass per counter set); counters of the LAST insert, summed over its dispatches\n" % (leg, $N))
        for k, e in out[leg]["kernels"].items():
            fo.write(k + "\n")
            for c, v in e.items():
                fo.write("    %-28s %.6g\n" % (c, v))
    for f in glob.glob("$OUT/%s_trace/**/*kernel_stats.csv" % leg, recursive=True):
        open("$OUT/%s_kernel_stats.csv" % leg, "w").write(open(f).read())
json.dump(out, open("$OUT/side_counters.json", "w"), indent=1)
for leg in out:
    print(leg, {k: out[leg][k] for k in ("hbm_bytes", "valu_insts", "lds_insts", "waves")})
PY
