cd $GRAFT_REPO_ROOT
echo "== sweep default (x4 from 1024 workgroups)"; timeout -k 10 200 python3 tools/exp/sweep_prof.py
echo "== sweep DGG_X4=0 (x5 everywhere)"; DGG_X4=0 timeout -k 10 200 python3 tools/exp/sweep_prof.py
echo "== sweep DGG_X4=0 DGG_X5=0 (x3)"; DGG_X4=0 DGG_X5=0 timeout -k 10 200 python3 tools/exp/sweep_prof.py
X5_CHILD=1 timeout -k 10 200 python3 tools/exp/x5_time.py time | head -3
