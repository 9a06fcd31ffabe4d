#!/bin/bash
# sub-results of bench.py (the serial steps) under settings of one environment switch.  usage: r6_subs.sh VAR reps v1 v2 ...
cd $GRAFT_REPO_ROOT
VAR=$1; REPS=$2; shift 2
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    export $VAR=$v
    python3 bench.py --no-cpu-baseline --no-other-mode --repeats 3 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);s=d['sub_results'];print('$VAR=$v', d['ms_per_step'], {k:v.get('ms_per_step') for k,v in s.items() if k in ('config2_cgan_step_b256','config3_eva_step_b1024','cgan_step_b1024','api_path_b1024','wgan_gp_step_b1024')}, s['config4_front_end_b512']['ms_per_step_without_encode'])"
  done
done
