#!/bin/bash
cd $GRAFT_REPO_ROOT
for args in "--depth 3" "--depth 4" "--rays 1000000 --resolution 0.05 --depth 3" "--rays 1000000 --resolution 0.05 --depth 4"; do
  python bench.py --steps 20 --warmup 3 --no-cpu $args 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; e=d.get('end_to_end',{})
print('$args', 'U=%d P=%.3g value=%.3e kernel_ms=%.4f step_ms=%.4f frac=%.4f e2e_ms=%.3f e2e_U=%d' % (d['config']['voxel_updates_per_scan'], d['config']['pair_evals_per_scan'], d['value'], r['kernel_ms'], d['ms_per_step'], r['frac'], e.get('ms_per_insert',-1), e.get('voxel_updates_per_scan',-1)), e.get('stages_s'))"
done
