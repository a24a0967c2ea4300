#!/bin/bash
# kernel timeline of the last scan of the 12-scan BGK-LV sequence (configs[3]): tools/prof/prof_lv_timeline.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lv_tl; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/lv_one.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
res, depth = 0.05, 5
params = dict(la3dm_amd.LV_YAML, resolution=res, block_depth=depth)
scans = [la3dm_amd.load_pcd(f"tests/golden/data/sim_unstructured/sim_unstructured_{i}.pcd") for i in range(1, 13)]
for rep in range(2):
    m = la3dm_amd.BGKLVOctoMap(**params, device=0)
    ts = []
    for xyz, origin in scans:
        t1 = time.perf_counter(); m.insert_pointcloud(xyz, origin, res, 0.1, 8.0); ts.append(time.perf_counter() - t1)
    print("per scan ms", [round(t * 1e3, 3) for t in ts], flush=True)
    time.sleep(0.01)
PY
rocprofv3 --output-format csv --kernel-trace -d $OUT -o t -- python /tmp/lv_one.py > $OUT/log.txt 2>&1
python tools/prof/timeline.py $OUT/t_kernel_trace.csv > $OUT/timeline.txt
tail -2 $OUT/log.txt | head -1
