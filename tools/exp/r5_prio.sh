# the cGAN chain on a high-priority stream (DGG_MAIN_PRIORITY=1), re-measured on round 5's kernels: joint step alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  python3 tools/exp/step_prof.py joint 1024 300 2>&1 | grep -v amdgpu.ids | sed 's/^/default  /'
  DGG_MAIN_PRIORITY=1 python3 tools/exp/step_prof.py joint 1024 300 2>&1 | grep -v amdgpu.ids | sed 's/^/priority /'
done
python3 tools/exp/step_prof.py joint 512 300 2>&1 | grep -v amdgpu.ids | sed 's/^/default  /'
DGG_MAIN_PRIORITY=1 python3 tools/exp/step_prof.py joint 512 300 2>&1 | grep -v amdgpu.ids | sed 's/^/priority /'
