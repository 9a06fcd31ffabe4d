cd $GRAFT_REPO_ROOT
X5_CHILD=1 timeout -k 10 200 python3 tools/exp/x5_time.py time 2>&1 | head -4
timeout -k 10 600 python -m pytest tests/test_gpu_steps.py -x -q -k "bf16" 2>&1 | tail -3
for v in 1 2; do timeout -k 10 120 python3 tools/exp/step_prof.py joint 1024 200; done
DGG_GEMM_ALGO=bf16 timeout -k 10 120 python3 tools/exp/step_prof.py eva 1024 200
timeout -k 10 120 python3 tools/exp/step_prof.py eva 1024 200
