cd $GRAFT_REPO_ROOT
for v in 1 2 3; do
DGG_X5_CUT=0 timeout -k 10 120 python3 tools/exp/step_prof.py joint 1024 200
DGG_X5_CUT=1 timeout -k 10 120 python3 tools/exp/step_prof.py joint 1024 200
done
DGG_X5_CUT=0 timeout -k 10 120 python3 tools/exp/step_prof.py joint 512 200
DGG_X5_CUT=1 timeout -k 10 120 python3 tools/exp/step_prof.py joint 512 200
