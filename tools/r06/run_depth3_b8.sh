#!/bin/bash
# A/B of the batch size of devmap_depth3.h: LA3DM_DEPTH3 = 4 | 8 (parity of the 8-block form first)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06/depth3; mkdir -p $O
LA3DM_DEPTH3=8 timeout 300 python -m pytest -q -m gpu -x tests/test_devmap_gpu.py -k "sequence_with_pruning or long_term or randomised or growing or config5 or moving" \
   tests/test_sharded_insert_gpu.py 2>&1 | tail -4 | tee $O/tests_b8.txt
for v in 4 8; do
  LA3DM_DEPTH3=$v timeout 200 python bench.py --gpus 1 --mode shard --steps 20 --warmup 3 > $O/shard_b$v.json 2> $O/shard_b$v.err
  LA3DM_DEPTH3=$v bash tools/prof/prof_devmap.sh 1000000 4 0.05 > /dev/null 2>&1; cp gpurun_out/prof_devmap/timeline.txt $O/timeline_1M_b$v.txt
  LA3DM_DEPTH3=$v bash tools/prof/prof_devmap.sh 200000 5 0.1 > /dev/null 2>&1; cp gpurun_out/prof_devmap/timeline.txt $O/timeline_200k_b$v.txt
done
python - <<PY
import json
for v in (4, 8):
    s = json.loads([l for l in open("$O/shard_b%d.json" % v) if l.startswith("{")][-1])
    print("LA3DM_DEPTH3=%d  configs[4] insert %.4f ms" % (v, s["ms_per_step"]))
PY
for f in $O/timeline_*_b?.txt; do echo $f; grep -E "dm_leaves|dm_commit_prune" $f | tail -3 | cut -c1-110; done
