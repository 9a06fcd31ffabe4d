# A / B: the dominant launches alone (x5_time.py prints HIP-event times of the step's large shapes) and the joint step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BASE=$GRAFT_REPO_ROOT/tools/exp/_ab/libdgg_base.so
for rep in 1 2; do
  DGG_LIB=$BASE X5_CHILD=1 python3 tools/exp/x5_time.py time 2>&1 | grep "wgrad TN\|forward NT  \|dgrad NN\|ragged TN" | sed 's/^/base /'
  X5_CHILD=1 python3 tools/exp/x5_time.py time 2>&1 | grep "wgrad TN\|forward NT  \|dgrad NN\|ragged TN" | sed 's/^/new  /'
done
for rep in 1 2 3; do
  DGG_LIB=$BASE python3 tools/exp/step_prof.py joint 1024 300 | sed 's/^/base /'
  python3 tools/exp/step_prof.py joint 1024 300 | sed 's/^/new  /'
done
DGG_LIB=$BASE python3 tools/exp/step_prof.py eva 1024 300 | sed 's/^/base /'
python3 tools/exp/step_prof.py eva 1024 300 | sed 's/^/new  /'
