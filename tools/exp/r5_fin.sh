# the forward products' slab reduce inside the launch (FwdFin) against the separate reduce launch (DGG_FWD_FIN=0): one call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  DGG_FWD_FIN=0 python3 tools/exp/step_prof.py joint 1024 300 2>&1 | grep -v amdgpu.ids | sed 's/^/separate /'
  python3 tools/exp/step_prof.py joint 1024 300 2>&1 | grep -v amdgpu.ids | sed 's/^/inside   /'
done
for w in "cgan 1024" "eva 1024" "joint 512" "cgan 256"; do
  DGG_FWD_FIN=0 python3 tools/exp/step_prof.py $w 300 2>&1 | grep -v amdgpu.ids | sed 's/^/separate /'
  python3 tools/exp/step_prof.py $w 300 2>&1 | grep -v amdgpu.ids | sed 's/^/inside   /'
done
