cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_steps.py -x -q 2>&1 | tail -2
for v in 1 2; do timeout -k 10 120 python3 tools/exp/step_prof.py joint 1024 200; done
DGG_SIDE_WORK=0 timeout -k 10 120 python3 tools/exp/step_prof.py joint 1024 200
DGG_SIDE_DIV=3 timeout -k 10 120 python3 tools/exp/step_prof.py joint 1024 200
DGG_SPLITK_TARGET=224 timeout -k 10 120 python3 tools/exp/step_prof.py joint 1024 200
DGG_SPLITK_TARGET=160 timeout -k 10 120 python3 tools/exp/step_prof.py joint 1024 200
timeout -k 10 120 python3 tools/exp/step_prof.py joint 512 200
timeout -k 10 120 python3 tools/exp/step_prof.py cgan 256 200
