# the fused fc2 stage: one 32 x 32 tile per workgroup, waves split K (fc2_stage.hip) against gemm_f32_kernel's 64 x 64 instantiation
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  DGG_FC2_TILE32=0 python3 tools/exp/step_prof.py joint 1024 300 2>&1 | grep -v amdgpu.ids | sed 's/^/64x64 /'
  python3 tools/exp/step_prof.py joint 1024 300 2>&1 | grep -v amdgpu.ids | sed 's/^/32x32 /'
done
for w in "cgan 1024" "eva 1024" "joint 512" "cgan 256" "cgan 256"; do
  DGG_FC2_TILE32=0 python3 tools/exp/step_prof.py $w 300 2>&1 | grep -v amdgpu.ids | sed 's/^/64x64 /'
  python3 tools/exp/step_prof.py $w 300 2>&1 | grep -v amdgpu.ids | sed 's/^/32x32 /'
done
