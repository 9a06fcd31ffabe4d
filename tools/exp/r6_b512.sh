#!/bin/bash
# the B = 512 joint step (config 4 without its encode) under settings of one environment switch.  usage: r6_b512.sh VAR reps v1 v2 ...
cd $GRAFT_REPO_ROOT
VAR=$1; REPS=$2; shift 2
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    export $VAR=$v
    python3 bench.py --no-cpu-baseline --no-other-mode --repeats 3 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);c=d['sub_results']['config4_front_end_b512'];print('$VAR=$v', c['ms_per_step_without_encode'], c['ms_per_step_with_encode'], d['sub_results']['config2_cgan_step_b256']['ms_per_step'])"
  done
done
