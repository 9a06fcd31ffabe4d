#!/bin/bash
# Round 6: the round's engine-side changes all OFF (XShare, deferred weight gradients, fine-grained backward; split optimiser
# step ON, look-ahead pass at half the workgroups, no forward pre-products, whole-grid evaluator Adam as in round 5) against the defaults, alternating on one box: bench.py headline + the serial sub-results.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in off on; do
    if [ $v = off ]; then export DGG_XSHARE=0 DGG_DEFER_WGRAD=0 DGG_FINE_BWD=0 DGG_SPLIT_ADAM=1 DGG_SIDE_PRE_LEVEL=1 DGG_PRE_FWD=0 DGG_ADAM_SIDE_WGS=2048; else unset DGG_XSHARE DGG_DEFER_WGRAD DGG_FINE_BWD DGG_SPLIT_ADAM DGG_SIDE_PRE_LEVEL DGG_PRE_FWD DGG_ADAM_SIDE_WGS; fi
    python3 bench.py --no-cpu-baseline --no-other-mode --repeats 5 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split('\n')[-1]);s=d['sub_results'];print('round6 changes $v', d['ms_per_step'], {k:v.get('ms_per_step') for k,v in s.items() if k in ('config2_cgan_step_b256','config3_eva_step_b1024','cgan_step_b1024','api_path_b1024','wgan_gp_step_b1024')}, s['config4_front_end_b512']['ms_per_step_with_encode'])"
  done
done
