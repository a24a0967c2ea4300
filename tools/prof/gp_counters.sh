#!/bin/bash
# Instruction counts of the GP predict + fuse launches per step, for the kernel source of THIS build, stamped:
#   profiles/gp_counters.json  (what bench.py's gp legs quote as roofline.valu_issue)
# usage (GPU box): bash tools/prof/gp_counters.sh  ->  gpurun_out/$ROUND/gp_counters/gp_counters.json, copy into profiles/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ROUND=${ROUND:-r06}   # output directory under gpurun_out/ and profiles/; the entries' "round" field
OUT=$GRAFT_REPO_ROOT/gpurun_out/$ROUND/gp_counters; rm -rf $OUT; mkdir -p $OUT
for D in 3 4; do
  timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_MFMA -d $OUT/d$D -o p -- \
    python bench.py --workload gp --depth $D --steps 1 --warmup 1 --no-cpu > $OUT/log$D.txt 2>&1 < /dev/null
done
python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
out = {}
for D in (3, 4):
    tot = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    for f in glob.glob("$OUT/d%d/**/*counter_collection.csv" % D, recursive=True):
        for row in csv.DictReader(open(f)):
            if "gp_predict_fuse" not in row["Kernel_Name"] or "eigen" in row["Kernel_Name"]:
                continue     # (the depth-3 leg also times option gp_mode 1 — gp_predict_fuse_eigen_kernel: not part of the default step)
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            disp[row["Kernel_Name"][:40]].add(row["Dispatch_Id"])
    steps = 2.0   # --steps 1 --warmup 1
    out["gp_rays50000_d%d" % D] = {
        "kernel": "gp_predict_fuse_small_kernel (size classes) + gp_predict_fuse_kernel, summed per step", "round": int("$ROUND"[1:]),
        "kernel_sha": bench.kernel_source_hash(("gp_kernels.h",)),
        "source": "profiles/gp_counters.json (tools/prof/gp_counters.sh: rocprofv3 --pmc, one pass)",
        "dispatches_per_step": {k: len(v) / steps for k, v in disp.items()},
        "valu_insts_per_launch": tot["SQ_INSTS_VALU"] / steps, "salu_insts_per_launch": tot["SQ_INSTS_SALU"] / steps,
        "lds_insts_per_launch": tot["SQ_INSTS_LDS"] / steps, "valu_active_units_per_launch": tot["SQ_ACTIVE_INST_VALU"] / steps,
        "f64_fma_per_launch": tot["SQ_INSTS_VALU_FMA_F64"] / steps, "mfma_per_launch": tot["SQ_INSTS_MFMA"] / steps,
        "waves_per_launch": tot["SQ_WAVES"] / steps}
    print(D, out["gp_rays50000_d%d" % D])
json.dump(out, open("$OUT/gp_counters.json", "w"), indent=1)
PY
