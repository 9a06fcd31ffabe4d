#!/bin/bash
# A/B/C... of an environment switch on one box: cycles through the settings, bench.py headline only.  usage: r6_abn.sh VAR reps v1 v2 ...
cd $GRAFT_REPO_ROOT
VAR=$1; REPS=$2; shift 2
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    export $VAR=$v
    python3 bench.py --no-cpu-baseline --no-sub --no-other-mode --repeats 7 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);print('$VAR=$v', d['ms_per_step'], d['config']['block_ms_per_step_min_median_max'])"
  done
done
