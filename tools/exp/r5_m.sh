cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_train_loop.py -x -q -k "benched_joint or train or interleaved or pipelining" 2>&1 | tail -2
for v in 1 2 3; do
DGG_LOSS_COPY_SIDE=0 timeout -k 10 120 python3 tools/exp/step_prof.py joint 1024 200
DGG_LOSS_COPY_SIDE=1 timeout -k 10 120 python3 tools/exp/step_prof.py joint 1024 200
done
DGG_LOSS_COPY_SIDE=0 timeout -k 10 120 python3 tools/exp/step_prof.py joint 512 200
DGG_LOSS_COPY_SIDE=1 timeout -k 10 120 python3 tools/exp/step_prof.py joint 512 200
