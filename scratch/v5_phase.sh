#!/bin/bash
# kernel time and per-tile instruction counts of bgk_predict_fuse_v5 by phase (ablate: 1 skip C, 2 skip B.., 4 skip D)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --steps 30 --warmup 5 --no-cpu --no-e2e "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'kernel_ms', round(d['roofline']['kernel_ms'],4))"; }
run
run --ablate 1
run --ablate 4
run --ablate 5
run --ablate 2
bash scratch/pmc_quick.sh
bash scratch/pmc_quick.sh --ablate 1
bash scratch/pmc_quick.sh --ablate 5
bash scratch/pmc_quick.sh --ablate 2
