#!/bin/bash
# One-rank process group over RCCL (DGG_FORCE_DISTRIBUTED=1): what the collective path costs a step before any byte moves.
run() { echo "== $*"; env "$@" MASTER_PORT=29533 timeout -k 10 150 python bench.py --no-sub --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['phase_us_after_step_begin'])"; }
run DGG_X=0
run DGG_FORCE_DISTRIBUTED=1
run DGG_FORCE_DISTRIBUTED=1 DGG_BUCKETED_AR=0
