#!/bin/bash
# A/B of an environment switch on one box: alternates the two settings, bench.py headline only.  usage: r6_ab.sh VAR A B [reps]
cd $GRAFT_REPO_ROOT
VAR=$1; A=$2; B=$3; REPS=${4:-3}
for rep in $(seq 1 $REPS); do
  for v in $A $B; do
    export $VAR=$v
    python3 bench.py --no-cpu-baseline --no-sub --no-other-mode --repeats 7 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);print('$VAR=$v', d['ms_per_step'], d['config']['block_ms_per_step_min_median_max'])"
  done
done
