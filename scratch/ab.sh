#!/bin/bash
# A/B of kernel options: each line = bench args
cd $GRAFT_REPO_ROOT
while read -r line; do
  [ -z "$line" ] && continue
  python bench.py --steps 30 --warmup 3 --no-cpu $line 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-40s kernel_ms=%.4f step_ms=%.4f frac=%.4f' % ('$line', r['kernel_ms'], d['ms_per_step'], r['frac']))"
done
