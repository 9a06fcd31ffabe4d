# A / B (base = tools/exp/_ab/libdgg_base.so): joint / cgan / eva steps alternating, 300 steps each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BASE=$GRAFT_REPO_ROOT/tools/exp/_ab/libdgg_base.so
for rep in 1 2 3 4; do
  DGG_LIB=$BASE python3 tools/exp/step_prof.py joint 1024 300 2>&1 | grep -v amdgpu.ids | sed 's/^/base /'
  python3 tools/exp/step_prof.py joint 1024 300 2>&1 | grep -v amdgpu.ids | sed 's/^/new  /'
done
for w in "cgan 1024" "eva 1024" "joint 512" "cgan 256"; do
  DGG_LIB=$BASE python3 tools/exp/step_prof.py $w 300 2>&1 | grep -v amdgpu.ids | sed 's/^/base /'
  python3 tools/exp/step_prof.py $w 300 2>&1 | grep -v amdgpu.ids | sed 's/^/new  /'
done
