#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --no-cpu-baseline --no-sub --no-other-mode --repeats 5 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);print('$1', d['ms_per_step'], d['side_streams'], d['config']['phase_us_after_step_begin'])"; }
for rep in 1 2; do
  unset DGG_SIDE_LOW; run default
  export DGG_SIDE_LOW=1; run low
done
