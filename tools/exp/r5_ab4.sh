# A / B (base = tools/exp/_ab/libdgg_base.so): the joint step only, N alternations of 300 steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BASE=$GRAFT_REPO_ROOT/tools/exp/_ab/libdgg_base.so
for rep in 1 2 3 4 5 6 7 8; do
  DGG_LIB=$BASE python3 tools/exp/step_prof.py joint 1024 300 2>&1 | grep -v amdgpu.ids | sed 's/^/base /'
  python3 tools/exp/step_prof.py joint 1024 300 2>&1 | grep -v amdgpu.ids | sed 's/^/new  /'
done
