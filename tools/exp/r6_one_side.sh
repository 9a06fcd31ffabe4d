#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --no-cpu-baseline --no-sub --no-other-mode --repeats 5 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);print('$1', d['ms_per_step'], d['config']['phase_us_after_step_begin'])"; }
for rep in 1 2; do
  unset DGG_ONE_SIDE DGG_SIDE_EVA_LEVEL DGG_SIDE_PRE_LEVEL; run default
  export DGG_ONE_SIDE=1
  for lv in "0 0" "1 1" "1 0" "0 1" "1 2"; do set -- $lv; export DGG_SIDE_EVA_LEVEL=$1 DGG_SIDE_PRE_LEVEL=$2; run "one_side eva=$1 pre=$2"; done
done
