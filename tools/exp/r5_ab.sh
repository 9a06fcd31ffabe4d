# A / B of the working tree's library against tools/exp/_ab/libdgg_base.so (tools/exp/build_base.sh), alternating, one call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BASE=$GRAFT_REPO_ROOT/tools/exp/_ab/libdgg_base.so
for rep in 1 2 3; do
  DGG_LIB=$BASE python3 tools/exp/step_prof.py joint 1024 300 | sed 's/^/base /'
  python3 tools/exp/step_prof.py joint 1024 300 | sed 's/^/new  /'
done
DGG_LIB=$BASE python3 tools/exp/step_prof.py cgan 1024 300 | sed 's/^/base /'
python3 tools/exp/step_prof.py cgan 1024 300 | sed 's/^/new  /'
DGG_LIB=$BASE python3 tools/exp/step_prof.py joint 512 300 | sed 's/^/base /'
python3 tools/exp/step_prof.py joint 512 300 | sed 's/^/new  /'
DGG_LIB=$BASE python3 tools/exp/step_prof.py cgan 256 300 | sed 's/^/base /'
python3 tools/exp/step_prof.py cgan 256 300 | sed 's/^/new  /'
