cd $GRAFT_REPO_ROOT
for v in 1 2 3; do
DGG_X5_WIDE=0 timeout -k 10 120 python3 tools/exp/step_prof.py joint 1024 200
DGG_X5_WIDE=1 timeout -k 10 120 python3 tools/exp/step_prof.py joint 1024 200
done
DGG_X5_WIDE=0 X5_CHILD=1 timeout -k 10 200 python3 tools/exp/x5_time.py time 2>/dev/null | head -2
DGG_X5_WIDE=1 X5_CHILD=1 timeout -k 10 200 python3 tools/exp/x5_time.py time 2>/dev/null | head -2
DGG_X5_WIDE=0 timeout -k 10 120 python3 tools/exp/step_prof.py eva 1024 200
DGG_X5_WIDE=1 timeout -k 10 120 python3 tools/exp/step_prof.py eva 1024 200
