#!/bin/bash
# phase times of the interleaved step under settings of one environment switch.  usage: r6_phases.sh VAR v1 v2 ...
cd $GRAFT_REPO_ROOT
VAR=$1; shift 1
for v in "$@"; do
  export $VAR=$v
  python3 bench.py --no-cpu-baseline --no-sub --no-other-mode --repeats 5 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);print('$VAR=$v', d['ms_per_step'], d['config']['phase_us_after_step_begin'])"
done
