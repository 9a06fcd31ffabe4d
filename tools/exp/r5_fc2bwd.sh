# fc2's backward stage in one launch (fc2_bwd_stage_kernel) against split-K GEMM + act_bwd (DGG_FC2_BWD_STAGE=0): one call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  DGG_FC2_BWD_STAGE=0 python3 tools/exp/step_prof.py joint 1024 300 2>&1 | grep -v amdgpu.ids | sed 's/^/two launches /'
  python3 tools/exp/step_prof.py joint 1024 300 2>&1 | grep -v amdgpu.ids | sed 's/^/one launch   /'
done
for w in "cgan 1024" "eva 1024" "joint 512" "cgan 256"; do
  DGG_FC2_BWD_STAGE=0 python3 tools/exp/step_prof.py $w 300 2>&1 | grep -v amdgpu.ids | sed 's/^/two launches /'
  python3 tools/exp/step_prof.py $w 300 2>&1 | grep -v amdgpu.ids | sed 's/^/one launch   /'
done
