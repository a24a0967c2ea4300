#!/bin/bash
# timing + instruction counts per tile of the BGK kernel variants (bench args after the tag)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --steps 30 --warmup 5 --no-cpu --no-e2e "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'kernel_ms', round(d['roofline']['kernel_ms'],4))"; }
run --variant 5
run --variant 0 --fifo-rows 8
run --variant 0 --fifo-rows 12
run --variant 0 --fifo-rows 16
run --variant 0 --fifo-rows 8 --ablate 1
run --variant 0 --fifo-rows 8 --ablate 4
run --variant 0 --fifo-rows 8 --ablate 5
run --variant 0 --fifo-rows 8 --ablate 2
bash scratch/pmc_quick.sh --variant 5
bash scratch/pmc_quick.sh --variant 0 --fifo-rows 8
bash scratch/pmc_quick.sh --variant 0 --fifo-rows 16
